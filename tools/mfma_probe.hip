// MFMA issue-rate probe: v_mfma_f32_32x32x2_f32 with 1/2/4 independent accumulator chains per wave,
// 1..4 waves per SIMD, all 256 CUs.  Prints achieved TFLOP/s and cycles per MFMA per SIMD (at 2.4 GHz).
// build: hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
static int run(int blocks_per_cu, float* out) {
    const int iters = 4096 / CH;
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_wave = (double)iters * 8 * CH;
    const double flops = mfma_per_wave * 4096.0 * grid * 4;
    const double per_simd = mfma_per_wave * blocks_per_cu;   // MFMAs a SIMD executes
    printf("chains %d, waves/SIMD %d: %7.1f us  %6.1f TFLOP/s  %.1f cycles per MFMA per SIMD @2.4GHz\n", CH, blocks_per_cu,
           ms * 1e3, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / per_simd);
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int w : {1, 2, 4}) { if (run<1>(w, out)) return 1; if (run<2>(w, out)) return 1; if (run<4>(w, out)) return 1; }
    return 0;
}
