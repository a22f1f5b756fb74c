// Tuning probe: the implicit-GEMM kernel compiled with per-K-step time stamps.
// build: hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off tools/igemm_prof.hip -o tools/igemm_prof.so
#define ICAP_IGEMM_PROF 1
#include <cstdarg>
#include <cstdio>
#include "../image_captioning_amd/csrc/conv_igemm.hip"
namespace icap {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
}
extern "C" {
int ig_linear(const float* x, const float* w, const float* b, float* y, int M, int K, int N) {
    return icap::conv_igemm_f32(x, w, nullptr, b, nullptr, nullptr, y, M, 1, 1, K, N, 1, 1, 1, 0, 0, 1, 1, 0, nullptr);
}
int ig_read(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(icap::g_igemm_prof), sizeof(unsigned long long) * 64 * 32 * 4);
}
}
