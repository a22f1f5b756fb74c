// How fast does one SIMD issue fp32 vector FMAs?  Independent v_fma_f32 / v_pk_fma_f32 chains, 1, 2 and 4 waves per SIMD, one
// workgroup per CU: cycles per instruction per SIMD = elapsed cycles / (instructions per wave x waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/valu_rate_probe && tools/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void probe(float* out, int iters, float a, float b) {
    float r[16];
    f2 p[8];
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) p[i] = f2{r[2 * i], r[2 * i + 1]};
    const f2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.9999f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (PK) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += r[i];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(out + (1 << 20)), (unsigned long long)(t1 - t0));
}

int main() {
    float* out;
    hipMalloc(&out, ((1 << 20) + 16) * sizeof(float));
    const int iters = 4096;
    for (int pk = 0; pk < 2; ++pk)
        for (int threads : {256, 512, 1024}) {
            hipMemset(out + (1 << 20), 0, 8);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            if (pk) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, out, 16, 1.0001f, 0.5f);
            else hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 0, 0, out, 16, 1.0001f, 0.5f);
            hipDeviceSynchronize();
            hipMemset(out + (1 << 20), 0, 8);
            hipEventRecord(e0);
            if (pk) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
            else hipLaunchKernelGGL(probe<0>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long cyc;
            hipMemcpy(&cyc, out + (1 << 20), 8, hipMemcpyDeviceToHost);
            const double inst = 32.0 * iters, wps = threads / 256.0;
            printf("%s  %d waves/SIMD: %.3f ms, %llu cycles (s_memtime) -> %.2f cycles per instruction per wave, %.2f per SIMD; %.1f TFLOP/s\n",
                   pk ? "v_pk_fma_f32" : "v_fma_f32   ", (int)wps, ms, cyc, cyc / inst, cyc / (inst * wps),
                   (pk ? 4.0 : 2.0) * 64 * inst * (threads / 64) * 256 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
