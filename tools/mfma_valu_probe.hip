// One wave's stream: v_mfma_f32_32x32x16_bf16 followed by NF independent v_fma_f32, repeated.  Cycles per (MFMA + NF FMAs)
// group with 1 and 2 waves per SIMD, accumulators in the accumulator registers ("a") or in ordinary registers ("v").
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_probe.hip -o tools/mfma_valu_probe && tools/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

template <int NF, int AGPR, int NACC, int PK = 0>
__global__ void probe(float* out, int iters, float a, float b) {
    f16v acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    b8 A, Bv;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)(threadIdx.x * 0.01f + i); Bv[i] = (__bf16)(i * 0.5f); }
    float r[24];
    for (int i = 0; i < 24; ++i) r[i] = threadIdx.x * 1e-3f + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p2[12];
    for (int i = 0; i < 12; ++i) p2[i] = f2{r[2 * i], r[2 * i + 1]};
    const f2 a2 = {a, a * 1.0001f}, b2 = {b, b * 0.9999f};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, acc[k % NACC], 0, 0, 0);
            if (AGPR) asm volatile("" : "+a"(acc[k % NACC])); else asm volatile("" : "+v"(acc[k % NACC]));
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2[(k * NF + i) % 12]) : "v"(a2), "v"(b2));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[(k * NF + i) % 24]) : "v"(a), "v"(b));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < NACC; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
    for (int i = 0; i < 24; ++i) s += r[i];
    for (int i = 0; i < 12; ++i) s += p2[i].x + p2[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(out + (1 << 20)), (unsigned long long)(t1 - t0));
}

template <int NF, int AGPR, int NACC, int PK = 0>
void run(float* out, int threads) {
    const int iters = 2048;
    hipLaunchKernelGGL((probe<NF, AGPR, NACC, PK>), dim3(256), dim3(threads), 0, 0, out, 8, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipMemset(out + (1 << 20), 0, 8);
    hipLaunchKernelGGL((probe<NF, AGPR, NACC, PK>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    unsigned long long cyc;
    hipMemcpy(&cyc, out + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("%s NF=%2d acc in %s x%d, %d wave(s)/SIMD: %6.1f cycles per group per wave, %6.1f per SIMD\n", PK ? "pk_fma" : "fma   ", NF, AGPR ? "AGPR" : "VGPR", NACC, threads / 256,
           cyc / (4.0 * iters), cyc / (4.0 * iters) / (threads / 256));
}

int main() {
    float* out;
    hipMalloc(&out, ((1 << 20) + 16) * sizeof(float));
    for (int threads : {256, 512}) {
        run<0, 1, 2>(out, threads); run<4, 1, 2>(out, threads); run<8, 1, 2>(out, threads); run<13, 1, 2>(out, threads); run<20, 1, 2>(out, threads);
        run<0, 0, 2>(out, threads); run<8, 0, 2>(out, threads); run<13, 0, 2>(out, threads); run<20, 0, 2>(out, threads);
        run<4, 1, 2, 1>(out, threads); run<7, 1, 2, 1>(out, threads); run<10, 1, 2, 1>(out, threads); run<10, 0, 2, 1>(out, threads);
    }
    return 0;
}
