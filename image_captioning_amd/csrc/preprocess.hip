// Image preprocessing on the device: the deterministic tail of the reference's image transform
// (data.py:170-172, datachar.py:129-131: Resize((224, 224)) on a PIL image, ToTensor, Normalize), bit-exact.
//
// torchvision resizes a PIL image with Pillow's BILINEAR resampler, which is not a two-tap interpolation: it is a
// separable triangle filter whose support grows with the down-scaling factor, evaluated in 22-bit fixed point, with the
// horizontally resized image rounded to 8 bits before the vertical pass.  The two passes are kept as two kernels with an
// 8-bit intermediate, because that rounding is part of the result.  The filter taps (integers) and tap ranges depend
// only on (input size, output size); the caller computes them once per size on the host and passes them in.
//
//   pass 1  tmp[b][y][xo]   = clip8((2^21 + sum_t src[b][y][bx[xo][0] + t][c] * kx[xo][t]) >> 22)   for c in r, g, b
//   pass 2  v               = clip8((2^21 + sum_t tmp[b][by[yo][0] + t][xo][c] * ky[yo][t]) >> 22)
//           out[b][c][yo][xo] = (v / 255 - mean[c]) / std[c]                                          (fp32, IEEE division)
//
// tmp packs r, g, b into one 32-bit word per pixel so the vertical pass reads one word per tap.
#include "common.hpp"

namespace icap {

constexpr int RS_PREC = 22;   // Pillow: PRECISION_BITS = 32 - 8 - 2

__device__ __forceinline__ unsigned rs_clip8(int v) {
    v >>= RS_PREC;            // arithmetic shift, as Pillow's lookup index
    return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one thread per (image row, output column); a workgroup covers 256 consecutive columns*rows of one image
__global__ __launch_bounds__(256) void resize_h_kernel(const unsigned char* __restrict__ src, const int* __restrict__ kx,
                                                       const int* __restrict__ bx, unsigned* __restrict__ tmp, int W, int Wo,
                                                       int ksx, long long rows) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Wo) return;
    const long long row = i / Wo;
    const int xo = (int)(i - row * Wo);
    int x0 = bx[2 * xo], n = bx[2 * xo + 1];
    n = n < 0 ? 0 : (n > ksx ? ksx : (n > W ? W : n));      // a malformed table must not turn into a wild read
    x0 = x0 < 0 ? 0 : (x0 > W - n ? W - n : x0);
    const unsigned char* s = src + (row * W + x0) * 3;
    const int* k = kx + (long long)xo * ksx;
    int r = 1 << (RS_PREC - 1), g = r, b = r;
    for (int t = 0; t < n; ++t) {
        const int w = k[t];
        r += (int)s[3 * t] * w;
        g += (int)s[3 * t + 1] * w;
        b += (int)s[3 * t + 2] * w;
    }
    tmp[i] = rs_clip8(r) | (rs_clip8(g) << 8) | (rs_clip8(b) << 16);
}

// one thread per output pixel; layout 0: out [B][3][Ho][Wo] (the reference's tensor), 1: out [B][Ho][Wo][4] with a zero
// fourth channel (the layout the encoder's first convolution reads)
__global__ __launch_bounds__(256) void resize_v_normalize_kernel(const unsigned* __restrict__ tmp, const int* __restrict__ ky,
                                                                 const int* __restrict__ by, const float* __restrict__ mean_std,
                                                                 float* __restrict__ out, int H, int Ho, int Wo, int ksy,
                                                                 int layout, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long long bimg = i / ((long long)Ho * Wo);
    const int rem = (int)(i - bimg * Ho * Wo);
    const int yo = rem / Wo, xo = rem - yo * Wo;
    int y0 = by[2 * yo], n = by[2 * yo + 1];
    n = n < 0 ? 0 : (n > ksy ? ksy : (n > H ? H : n));
    y0 = y0 < 0 ? 0 : (y0 > H - n ? H - n : y0);
    const unsigned* s = tmp + ((bimg * H + y0) * Wo + xo);
    const int* k = ky + (long long)yo * ksy;
    int r = 1 << (RS_PREC - 1), g = r, b = r;
    for (int t = 0; t < n; ++t) {
        const unsigned p = s[(long long)t * Wo];
        const int w = k[t];
        r += (int)(p & 255u) * w;
        g += (int)((p >> 8) & 255u) * w;
        b += (int)((p >> 16) & 255u) * w;
    }
    const float fr = ((float)rs_clip8(r) / 255.0f - mean_std[0]) / mean_std[3];
    const float fg = ((float)rs_clip8(g) / 255.0f - mean_std[1]) / mean_std[4];
    const float fb = ((float)rs_clip8(b) / 255.0f - mean_std[2]) / mean_std[5];
    if (layout == 0) {
        const long long plane = (long long)Ho * Wo;
        float* o = out + bimg * 3 * plane + rem;
        o[0] = fr;
        o[plane] = fg;
        o[2 * plane] = fb;
    } else {
        reinterpret_cast<float4*>(out)[i] = make_float4(fr, fg, fb, 0.f);
    }
}

// Fast forms used when the sizes allow (anything else takes the two kernels above).
//
// Horizontal pass, one workgroup per image row: the row's bytes are copied to LDS with coalesced 32-bit loads (the row
// starts at an arbitrary byte, so the copy starts at the aligned word below it), then each thread filters one or more
// output pixels from LDS.  rows = B * H; needs 3 * W + 7 bytes of LDS.
__global__ __launch_bounds__(256) void resize_h_row_kernel(const unsigned char* __restrict__ src, const int* __restrict__ kx,
                                                           const int* __restrict__ bx, unsigned* __restrict__ tmp, int W, int Wo,
                                                           int ksx, long long total_bytes) {
    extern __shared__ unsigned row_lds[];
    const long long row = blockIdx.x;
    const long long off = row * W * 3;
    const int shift = (int)(off & 3);
    const long long base = off - shift;
    const int nwords = (shift + 3 * W + 3) >> 2;
    const unsigned* src32 = reinterpret_cast<const unsigned*>(src + base);
    for (int i = threadIdx.x; i < nwords; i += 256) {
        unsigned v;
        if (base + 4LL * i + 4 <= total_bytes) {
            v = src32[i];
        } else {            // the last word of the whole buffer: never read past its end
            v = 0;
            for (int j = 0; j < 4; ++j)
                if (base + 4LL * i + j < total_bytes) v |= (unsigned)src[base + 4LL * i + j] << (8 * j);
        }
        row_lds[i] = v;
    }
    __syncthreads();
    const unsigned char* rb = reinterpret_cast<const unsigned char*>(row_lds) + shift;
    for (int xo = threadIdx.x; xo < Wo; xo += 256) {
        int x0 = bx[2 * xo], n = bx[2 * xo + 1];
        n = n < 0 ? 0 : (n > ksx ? ksx : (n > W ? W : n));
        x0 = x0 < 0 ? 0 : (x0 > W - n ? W - n : x0);
        const unsigned char* s = rb + 3 * x0;
        const int* k = kx + (long long)xo * ksx;
        int r = 1 << (RS_PREC - 1), g = r, b = r;
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
            r += (int)s[3 * t] * w;
            g += (int)s[3 * t + 1] * w;
            b += (int)s[3 * t + 2] * w;
        }
        tmp[row * Wo + xo] = rs_clip8(r) | (rs_clip8(g) << 8) | (rs_clip8(b) << 16);
    }
}

// Vertical pass + normalisation, four output pixels of one row per thread (Wo % 4 == 0): 16-byte loads of the packed
// intermediate, 16-byte stores.  grid = (Ho * Wo / 4 / 256 rounded up, B).
__global__ __launch_bounds__(256) void resize_v4_normalize_kernel(const unsigned* __restrict__ tmp, const int* __restrict__ ky,
                                                                  const int* __restrict__ by, const float* __restrict__ mean_std,
                                                                  float* __restrict__ out, int H, int Ho, int Wo, int ksy, int layout) {
    const int wq = Wo >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Ho * wq) return;
    const int yo = idx / wq, x4 = (idx - yo * wq) << 2;
    const long long bimg = blockIdx.y;
    int y0 = by[2 * yo], n = by[2 * yo + 1];
    n = n < 0 ? 0 : (n > ksy ? ksy : (n > H ? H : n));
    y0 = y0 < 0 ? 0 : (y0 > H - n ? H - n : y0);
    const unsigned* s = tmp + ((bimg * H + y0) * Wo + x4);
    const int* k = ky + (long long)yo * ksy;
    int acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 1 << (RS_PREC - 1);
    for (int t = 0; t < n; ++t) {
        const uint4 p = *reinterpret_cast<const uint4*>(s + (long long)t * Wo);
        const int w = k[t];
        const unsigned u[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j][0] += (int)(u[j] & 255u) * w;
            acc[j][1] += (int)((u[j] >> 8) & 255u) * w;
            acc[j][2] += (int)((u[j] >> 16) & 255u) * w;
        }
    }
    float f[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) f[j][c] = ((float)rs_clip8(acc[j][c]) / 255.0f - mean_std[c]) / mean_std[3 + c];
    if (layout == 0) {
        const long long plane = (long long)Ho * Wo;
        float* o = out + bimg * 3 * plane + (long long)yo * Wo + x4;
#pragma unroll
        for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(o + c * plane) = make_float4(f[0][c], f[1][c], f[2][c], f[3][c]);
    } else {
        float4* o = reinterpret_cast<float4*>(out) + (bimg * Ho + yo) * Wo + x4;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = make_float4(f[j][0], f[j][1], f[j][2], 0.f);
    }
}

long long resize_normalize_ws_bytes(int B, int H, int Wo) { return (long long)B * H * Wo * 4; }

// src [B][H][W][3] uint8; kx [Wo][ksx], bx [Wo][2], ky [Ho][ksy], by [Ho][2] int32; mean_std: 3 means then 3 stds;
// tmp: resize_normalize_ws_bytes(B, H, Wo) bytes.  Tap ranges are clamped to the image in the kernels.
int resize_normalize_u8_f32(const unsigned char* src, int B, int H, int W, const int* kx, const int* bx, int ksx, const int* ky,
                            const int* by, int ksy, int Ho, int Wo, const float* mean_std, unsigned* tmp, float* out, int layout,
                            hipStream_t st) {
    if (!src || !kx || !bx || !ky || !by || !mean_std || !tmp || !out || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 ||
        ksx <= 0 || ksy <= 0 || (layout != 0 && layout != 1)) {
        set_error("resize_normalize: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (layout == 1 && !aligned16(out)) { set_error("resize_normalize: out must be 16-byte aligned"); return ICAP_ERR_ALIGN; }
    const long long n1 = (long long)B * H * Wo, n2 = (long long)B * Ho * Wo;
    const long long g1 = (n1 + 255) / 256, g2 = (n2 + 255) / 256;
    if (g1 > 0x7fffffffLL || g2 > 0x7fffffffLL) { set_error("resize_normalize: too large"); return ICAP_ERR_LIMIT; }
    const long long rows = (long long)B * H;
    const size_t row_lds = (size_t)(3LL * W + 7) / 4 * 4 + 4;
    if (row_lds <= 48 * 1024 && rows <= 0x7fffffffLL && (reinterpret_cast<uintptr_t>(src) & 3) == 0)
        hipLaunchKernelGGL(resize_h_row_kernel, dim3((unsigned)rows), dim3(256), row_lds, st, src, kx, bx, tmp, W, Wo, ksx, rows * W * 3);
    else
        hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)g1), dim3(256), 0, st, src, kx, bx, tmp, W, Wo, ksx, rows);
    // the NCHW planes start at multiples of Ho * Wo floats: 16-byte stores need that product (and out) aligned as well
    if (Wo % 4 == 0 && B <= 65535 && aligned16(out) && aligned16(tmp))
        hipLaunchKernelGGL(resize_v4_normalize_kernel, dim3((unsigned)(((long long)Ho * (Wo / 4) + 255) / 256), (unsigned)B), dim3(256), 0,
                           st, tmp, ky, by, mean_std, out, H, Ho, Wo, ksy, layout);
    else
        hipLaunchKernelGGL(resize_v_normalize_kernel, dim3((unsigned)g2), dim3(256), 0, st, tmp, ky, by, mean_std, out, H, Ho, Wo, ksy,
                           layout, n2);
    return launch_status("resize_normalize");
}

// ------------------------------------------------------------------------------------------------------------------
// The random augmentations of the same transform for GIVEN per-image parameters (data.py:167-169, datachar.py:125-127:
// RandomHorizontalFlip, RandomRotation(20), ColorJitter(contrast=0.3) - on a PIL image, i.e. Pillow's
// transpose(FLIP_LEFT_RIGHT), rotate(angle, NEAREST, fillcolor=0) and ImageEnhance.Contrast), bit-exact; the draws
// themselves are the caller's.
//
//   pass 1  geometry: destination (x, y) takes the source pixel (sx, sy) = ((a2 + x*a0 + y*a1) >> 16, (a5 + x*a3 + y*a4) >> 16)
//           of the (optionally mirrored) image - Pillow's 16.16 fixed-point NEAREST affine pass, the caller supplies the
//           six integers - or 0 when the source lies outside; the luma (19595 r + 38470 g + 7471 b + 32768) >> 16 of
//           every result pixel is summed per image (integer atomics: the sum does not depend on the order).
//   pass 2  contrast: mean = int(sum / (H*W) + 0.5); v = mean + factor * (v - mean) in fp32 without contraction,
//           truncated for 0 <= factor <= 1, else clipped to [0, 255] and truncated (Pillow's ImagingBlend).
struct AugParams { int flip, a0, a1, a2, a3, a4, a5, pad; };

__global__ __launch_bounds__(256) void augment_geom_kernel(const unsigned char* __restrict__ src, const AugParams* __restrict__ prm,
                                                           unsigned char* __restrict__ dst, unsigned long long* __restrict__ sums,
                                                           int H, int W, int blocks_per_image) {
    const int b = blockIdx.x / blocks_per_image;
    const int i = (blockIdx.x % blocks_per_image) * 256 + threadIdx.x;
    const AugParams q = prm[b];
    unsigned luma = 0;
    if (i < H * W) {
        const int y = i / W, x = i - y * W;
        const long long fx = (long long)q.a2 + (long long)x * q.a0 + (long long)y * q.a1;
        const long long fy = (long long)q.a5 + (long long)x * q.a3 + (long long)y * q.a4;
        long long sx = fx >> 16;
        const long long sy = fy >> 16;
        unsigned r = 0, g = 0, bl = 0;
        if (sx >= 0 && sx < W && sy >= 0 && sy < H) {
            if (q.flip) sx = W - 1 - sx;
            const unsigned char* s = src + (((size_t)b * H + (size_t)sy) * W + (size_t)sx) * 3;
            r = s[0]; g = s[1]; bl = s[2];
        }
        unsigned char* d = dst + ((size_t)b * H * W + (size_t)i) * 3;
        d[0] = (unsigned char)r; d[1] = (unsigned char)g; d[2] = (unsigned char)bl;
        luma = (r * 19595u + g * 38470u + bl * 7471u + 0x8000u) >> 16;
    }
    __shared__ unsigned part[4];
    unsigned v = luma;
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[b], (unsigned long long)(part[0] + part[1] + part[2] + part[3]));
}

__global__ __launch_bounds__(256) void augment_contrast_kernel(unsigned char* __restrict__ img, const unsigned long long* __restrict__ sums,
                                                               const float* __restrict__ factor, long long per_image, long long pixels) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_image) return;
    const float mean = (float)(int)((double)sums[b] / (double)pixels + 0.5);
    const float f = factor[b];
    unsigned char* p = img + (size_t)b * per_image + i;
    const float t = mean + f * ((float)(int)*p - mean);
    int o;
    if (f >= 0.f && f <= 1.f) o = (int)t;
    else o = t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
    *p = (unsigned char)o;
}

long long augment_ws_bytes(int B) { return 8LL * B; }

// src, dst uint8 [B][H][W][3] (dst != src); params int32 [B][8] = flip, a0..a5, 0; contrast fp32 [B]; ws augment_ws_bytes(B).
int augment_u8(const unsigned char* src, int B, int H, int W, const int* params, const float* contrast, unsigned long long* ws,
               unsigned char* dst, hipStream_t st) {
    if (!src || !params || !contrast || !ws || !dst || B <= 0 || H <= 0 || W <= 0) { set_error("augment: bad arguments"); return ICAP_ERR_ARG; }
    if (src == dst) { set_error("augment: the geometric pass cannot run in place"); return ICAP_ERR_ARG; }
    if (H > 16384 || W > 16384 || B > 65535) { set_error("augment: image %dx%d or batch %d beyond 16384 x 16384 x 65535", H, W, B); return ICAP_ERR_LIMIT; }
    if (((uintptr_t)params & 3) || ((uintptr_t)ws & 7)) { set_error("augment: params / ws misaligned"); return ICAP_ERR_ALIGN; }
    hipError_t e = hipMemsetAsync(ws, 0, 8ULL * B, st);
    if (e != hipSuccess) { set_error("augment: memset: %s", hipGetErrorString(e)); return ICAP_ERR_HIP_BASE + (int)e; }
    const int bpi = (H * W + 255) / 256;
    hipLaunchKernelGGL(augment_geom_kernel, dim3((unsigned)(B * (long long)bpi)), dim3(256), 0, st, src, reinterpret_cast<const AugParams*>(params), dst, ws, H, W, bpi);
    const long long per = 3LL * H * W;
    hipLaunchKernelGGL(augment_contrast_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)B), dim3(256), 0, st, dst, ws, contrast, per, (long long)H * W);
    return launch_status("augment");
}

}  // namespace icap
