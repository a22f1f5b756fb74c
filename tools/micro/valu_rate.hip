// Issue cost of vector instructions on one SIMD: every workgroup is one wave, WPS waves per SIMD; each wave runs a long
// stream of independent instructions of one kind.  Prints cycles per instruction per SIMD.
// hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rate.hip -o tools/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int iters, float seed) {
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + i, seed - i};
    f2 m = {seed * 0.5f, seed * 0.25f};
    float sm = __builtin_amdgcn_readfirstlane(seed * 0.5f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(m.x), "v"(m.y)); }
                if (KIND == 1) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(m)); }
                if (KIND == 2) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(m), "s"((unsigned long long)__float_as_uint(sm))); }
                if (KIND == 3) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); }
                if (KIND == 4) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i].x)); }
                if (KIND == 5) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x)); }
                if (KIND == 6) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m)); }
                if (KIND == 7) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].x)); }
            }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, float* out) {
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int grid = 256 * 4 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, 10, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 32 * wps);
        printf("%-28s waves/SIMD %d: %.2f cycles per instruction per SIMD (at 2.4 GHz)\n", name, wps, cyc);
    }
}
int main() {
    float* out; hipMalloc(&out, 4096);
    run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<2>("v_pk_fma_f32 sgpr,op_sel", out); run<3>("v_pk_mul_f32", out);
    run<6>("v_pk_add_f32", out); run<5>("v_mul_f32", out); run<4>("v_exp_f32", out); run<7>("v_rcp_f32", out);
    return 0;
}
