// Latency floor probe for the small-batch decode path (build: hipcc -O3 --offload-arch=gfx950 latency_probe.hip -o latency_probe).
// Measures, per dependent kernel in one stream: an empty launch, a launch that streams 4 MB of weights,
// and the cost of one grid-wide barrier inside a persistent kernel (bounded spin, cannot hang).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_empty() {}

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ w, const float* __restrict__ x, float* __restrict__ y, int n4_per_block) {
    // every block: read x (small, produced by the previous kernel), then its slice of w, reduce, write one float
    float xv = x[threadIdx.x & 63];
    const float4* p = w + (size_t)blockIdx.x * n4_per_block;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n4_per_block; i += 256) { float4 v = p[i]; acc += (v.x + v.y + v.z + v.w) * xv; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) y[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

// sense-reversal barrier on one global counter; spin bounded so that a scheduling surprise ends in an error flag, not a hang
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned nblocks, unsigned& epoch, int* fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        epoch += 1;
        const unsigned target = epoch * nblocks;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 2000000) { *fail = 1; ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void k_barriers(unsigned* counter, int iters, int* fail, float* sink) {
    unsigned epoch = 0;
    float a = 0.f;
    for (int i = 0; i < iters; ++i) {
        a += 1.f;
        if (!grid_barrier(counter, gridDim.x, epoch, fail)) break;
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = a;
}

// barrier + a dependent exchange: every block writes a value, barrier, every block reads all values (like a GEMV step)
__global__ __launch_bounds__(256) void k_exchange(unsigned* counter, int iters, int* fail, float* buf, float* sink) {
    unsigned epoch = 0;
    float a = 1.f;
    const int n = gridDim.x;
    for (int i = 0; i < iters; ++i) {
        float* cur = buf + (size_t)(i & 1) * n;
        if (threadIdx.x == 0) __hip_atomic_store(cur + blockIdx.x, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!grid_barrier(counter, n, epoch, fail)) break;
        float s = 0.f;
        for (int j = threadIdx.x; j < n; j += 256) s += __hip_atomic_load(cur + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        a = s * (1.0f / n);
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = a;
}

template <typename F>
static float time_us(hipStream_t st, int reps, F&& f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int blocks = 512, n4 = 512;  // 512 blocks x 512 float4 = 4 MB
    float4* w;
    float *x, *y, *sink, *buf;
    unsigned* counter;
    int* fail;
    CK(hipMalloc(&w, (size_t)blocks * n4 * 16 * 8));
    CK(hipMemset(w, 0, (size_t)blocks * n4 * 16 * 8));
    CK(hipMalloc(&x, 4096 * 4));
    CK(hipMalloc(&y, 4096 * 4));
    CK(hipMemset(x, 0, 4096 * 4));
    CK(hipMemset(y, 0, 4096 * 4));
    CK(hipMalloc(&sink, 4096 * 4));
    CK(hipMalloc(&buf, 8192 * 4));
    CK(hipMalloc(&counter, 4));
    CK(hipMalloc(&fail, 4));
    CK(hipMemset(fail, 0, 4));

    printf("empty <<<1,64>>>       : %6.2f us/kernel\n", time_us(st, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }));
    printf("empty <<<512,256>>>    : %6.2f us/kernel\n", time_us(st, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, st); }));
    // ping-pong x/y so that each kernel depends on the previous one's output
    int flip = 0;
    printf("stream 4 MB, dependent : %6.2f us/kernel\n", time_us(st, 2000, [&] {
               hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, st, w, flip ? y : x, flip ? x : y, n4);
               flip ^= 1;
           }));
    printf("stream 16 MB, dependent: %6.2f us/kernel\n", time_us(st, 1000, [&] {
               hipLaunchKernelGGL(k_stream, dim3(blocks * 4), dim3(256), 0, st, w, flip ? y : x, flip ? x : y, n4);
               flip ^= 1;
           }));
    // HIP graph of 60 dependent 4 MB kernels
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 60; ++i) { hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, st, w, (i & 1) ? y : x, (i & 1) ? x : y, n4); }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float us = time_us(st, 50, [&] { CK(hipGraphLaunch(ge, st)); });
        printf("graph of 60 x 4 MB     : %6.2f us/kernel\n", us / 60);
    }
    for (int nb : {64, 128, 256, 512}) {
        const int iters = 2000;
        CK(hipMemset(counter, 0, 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_barriers, dim3(nb), dim3(256), 0, st, counter, iters, fail, sink);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        int hf = 0;
        CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
        printf("grid barrier, %3d blocks: %6.2f us/barrier%s\n", nb, ms * 1000.f / iters, hf ? "  (SPIN LIMIT HIT)" : "");
        if (hf) return 2;
        CK(hipMemset(counter, 0, 4));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k_exchange, dim3(nb), dim3(256), 0, st, counter, iters, fail, buf, sink);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
        printf("barrier + exchange, %3d : %6.2f us/step%s\n", nb, ms * 1000.f / iters, hf ? "  (SPIN LIMIT HIT)" : "");
        if (hf) return 2;
    }
    return 0;
}
