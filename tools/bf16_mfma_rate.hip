// How many cycles does a SIMD spend per v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 when nothing else is in the way?
// One or two waves per SIMD, 1..8 independent accumulators, no memory traffic.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH, bool BIG>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(threadIdx.x, 1, 2, 3)), b = __builtin_bit_cast(bf16x8, make_uint4(4, threadIdx.x, 6, 7));
    float r = 0.f;
    if constexpr (BIG) {
        f32x16 acc[CH];
        for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < CH; ++c) r += acc[c][0];
    } else {
        f32x4 acc[CH];
        for (int c = 0; c < CH; ++c) for (int e = 0; e < 4; ++e) acc[c][e] = 0.f;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < CH; ++c) r += acc[c][0];
    }
    if (r == 123.456f) out[0] = r;
}

template <int CH, bool BIG>
static void run(float* out, int threads) {
    const int iters = 4096 / CH * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CH, BIG>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<CH, BIG>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = (double)iters * CH * (threads / 256);
    const double flops = (BIG ? 32768.0 : 16384.0) * mfmas_per_simd * 1024;
    printf("%s chains %d waves/SIMD %d: %8.1f us  %7.1f TFLOP/s  %5.1f cycles per MFMA per SIMD @2.4GHz\n", BIG ? "32x32x16" : "16x16x32", CH,
           threads / 256, ms * 1e3, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / mfmas_per_simd);
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    run<1, true>(out, 256); run<2, true>(out, 256); run<4, true>(out, 256); run<8, true>(out, 256);
    run<1, true>(out, 512); run<4, true>(out, 512);
    run<1, false>(out, 256); run<2, false>(out, 256); run<4, false>(out, 256); run<8, false>(out, 256);
    run<4, false>(out, 512);
    return 0;
}
