// Do matrix-core and vector work of DIFFERENT waves on one SIMD overlap?  512-thread workgroups, one per CU: waves 0-3 (one per
// SIMD) run an MFMA loop, waves 4-7 (their SIMD partners) a VALU loop; each half is timed alone and together.
//   mfma: v_mfma_f32_32x32x16_bf16 (or the fp32 32x32x2) on 4 independent accumulators
//   valu: plain fma chains / a swish-like body (exp2 + rcp + 3 fma) on 8 independent registers
// Then one wave doing both in its own instruction stream (vector instructions placed after each MFMA).
// Prints microseconds; "sum" and "max" of the two alone times bracket the together time.
// build: hipcc -O3 --offload-arch=gfx950 coexec_probe.hip -o coexec_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool BF16>
__device__ __forceinline__ float mfma_body(int iters, float seed) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    if constexpr (BF16) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    } else {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed + 1.f, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    return s;
}

template <bool SWISH>
__device__ __forceinline__ float valu_body(int iters, float seed) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (SWISH) {
                    const float t = fmaf(v[j], 0.999f, 0.001f);
                    v[j] = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695f * t));
                } else {
                    v[j] = fmaf(v[j], 0.999f, 0.001f);
                    v[j] = fmaf(v[j], 1.001f, -0.001f);
                    v[j] = fmaf(v[j], 0.998f, 0.002f);
                    v[j] = fmaf(v[j], 1.002f, -0.002f);
                }
            }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += v[j];
    return s;
}

// One wave doing both in its own instruction stream: VPM vector instructions after every MFMA (independent of it)
template <bool BF16, int VPM>
__global__ __launch_bounds__(256) void k_mixed(float* out, int mode, int iters, float seed0) {
    const float seed = seed0 + threadIdx.x;
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (mode & 1) {
                    if constexpr (BF16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
                    else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, seed + 1.f, acc[c], 0, 0, 0);
                }
                if (mode & 2) {
#pragma unroll
                    for (int q = 0; q < VPM; ++q) v[(c * VPM + q) & 7] = fmaf(v[(c * VPM + q) & 7], 0.999f, 0.001f);
                }
            }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool BF16, int VPM>
static int run_mixed(float* out, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float t[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL((k_mixed<BF16, VPM>), dim3(256), dim3(256), 0, 0, out, mode, iters, 1.f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mixed<BF16, VPM>), dim3(256), dim3(256), 0, 0, out, mode, iters, 1.f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t[mode], e0, e1));
    }
    printf("%s mfma with %2d fma after each, ONE wave per SIMD: mfma only %7.1f us, fma only %7.1f us, interleaved %7.1f us  (max %7.1f, sum %7.1f)\n",
           BF16 ? "bf16 32x32x16" : "fp32 32x32x2 ", VPM, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, (t[1] > t[2] ? t[1] : t[2]) * 1e3, (t[1] + t[2]) * 1e3);
    return 0;
}

// mode bit 0: the MFMA half runs, bit 1: the VALU half runs.  SPLIT_WG: the role is chosen per workgroup (blockIdx parity)
template <bool BF16, bool SWISH, bool SPLIT_WG>
__global__ __launch_bounds__(SPLIT_WG ? 256 : 512) void k(float* out, int mode, int mi, int vi, float seed) {
    const bool mf = SPLIT_WG ? (blockIdx.x & 1) == 0 : threadIdx.x < 256;
    float s = 0.f;
    if (mode & 4) { if (mf) __builtin_amdgcn_s_setprio(3); }     // the matrix-core waves get issue priority
    if (mode & 8) { if (!mf) __builtin_amdgcn_s_setprio(3); }    // ... or the vector waves
    if (mf) { if (mode & 1) s = mfma_body<BF16>(mi, seed + threadIdx.x); }
    else if (mode & 2) s = valu_body<SWISH>(vi, seed + threadIdx.x);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool BF16, bool SWISH, bool SPLIT_WG>
static int run(float* out, int mi, int vi) {
    const int grid = SPLIT_WG ? 512 : 256, block = SPLIT_WG ? 256 : 512;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float t[12] = {0};
    for (int mode : {1, 2, 3, 7, 11}) {
        hipLaunchKernelGGL((k<BF16, SWISH, SPLIT_WG>), dim3(grid), dim3(block), 0, 0, out, mode, mi, vi, 1.f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<BF16, SWISH, SPLIT_WG>), dim3(grid), dim3(block), 0, 0, out, mode, mi, vi, 1.f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&t[mode], e0, e1));
    }
    printf("%s mfma | %s valu | %s: mfma alone %7.1f us, valu alone %7.1f us, together %7.1f us  (max %7.1f, sum %7.1f); mfma waves at priority 3: %7.1f, valu waves at priority 3: %7.1f\n",
           BF16 ? "bf16 32x32x16" : "fp32 32x32x2 ", SWISH ? "swish" : "fma  ", SPLIT_WG ? "two 256-thread workgroups per CU" : "one 512-thread workgroup per CU  ",
           t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, (t[1] > t[2] ? t[1] : t[2]) * 1e3, (t[1] + t[2]) * 1e3, t[7] * 1e3, t[11] * 1e3);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 512 * 512 * 4));
    // iteration counts chosen so that each half alone takes about the same time
    if (run<true, false, false>(out, 4000, 2000)) return 1;
    if (run<true, true, false>(out, 4000, 700)) return 1;
    if (run<false, false, false>(out, 2000, 2000)) return 1;
    if (run<false, true, false>(out, 2000, 700)) return 1;
    if (run_mixed<true, 4>(out, 4000)) return 1;
    if (run_mixed<true, 8>(out, 4000)) return 1;
    if (run_mixed<true, 12>(out, 4000)) return 1;
    if (run_mixed<false, 8>(out, 2000)) return 1;
    if (run_mixed<false, 16>(out, 2000)) return 1;
    if (run_mixed<false, 24>(out, 2000)) return 1;
    return 0;
}
