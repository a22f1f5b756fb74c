// Backward-pass kernels specific to the convolutional encoder (EfficientNet-B0 in training mode):
// depthwise convolution gradients, the squeeze-excite gate's per-image products, and the im2col used
// by the stem convolution's weight gradient.  Activations are NHWC fp32, viewed as [B*H*W][C] matrices.
#include "common.hpp"

namespace icap {

// ---------------------------------------------------------------- per-image channel affine
// y[b,p,c] = x[b,p,c] * g[b][c] + a[b][c] * alpha   (x == null: broadcast only; g == null: 1; a == null: 0).
// Forward: the squeeze-excite gate (g = sigmoid gate).  Backward: gradient through the gate plus the
// gradient of the pooled mean spread over the image (a = dpooled, alpha = 1/HW); global-average-pool backward.
__global__ __launch_bounds__(256) void chan_affine_kernel(const float4* __restrict__ x, const float4* __restrict__ g,
                                                          const float4* __restrict__ a, float alpha, int HW, int C4,
                                                          long long n4, float4* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const long long row = i / C4;
        const int c4 = (int)(i - row * C4);
        const int b = (int)(row / HW);
        float4 v = x ? x[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (g) { const float4 gv = g[(size_t)b * C4 + c4]; v.x *= gv.x; v.y *= gv.y; v.z *= gv.z; v.w *= gv.w; }
        if (a) { const float4 av = a[(size_t)b * C4 + c4]; v.x += av.x * alpha; v.y += av.y * alpha; v.z += av.z * alpha; v.w += av.w * alpha; }
        y[i] = v;
    }
}

int chan_affine_f32(const float* x, const float* g, const float* a, float alpha, int B, int HW, int C, float* y, hipStream_t st) {
    if (!y || (!x && !a) || B <= 0 || HW <= 0 || C <= 0) { set_error("chan_affine: bad arguments"); return ICAP_ERR_ARG; }
    if (C % 4 || !aligned16(y) || (x && !aligned16(x)) || (g && !aligned16(g)) || (a && !aligned16(a))) {
        set_error("chan_affine: C %% 4 and 16-byte alignment required");
        return ICAP_ERR_ALIGN;
    }
    const long long n4 = (long long)B * HW * (C / 4);
    long long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(chan_affine_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<const float4*>(g), reinterpret_cast<const float4*>(a), alpha, HW, C / 4, n4,
                       reinterpret_cast<float4*>(y));
    return launch_status("chan_affine");
}

// out[b][c] = scale * sum_p x[b,p,c] * y[b,p,c]  (y == null: sum of x).  TX threads span 4*TX adjacent channels
// (one float4 each), 256/TX threads share the rows of one chunk of one image; chunk partials go to
// part[b][chunk][C] and image_dot_finish_kernel adds them in order.
template <int TX>
__global__ __launch_bounds__(256) void image_dot_kernel(const float* __restrict__ x, const float* __restrict__ y, int HW, int C,
                                                        int rows_per_chunk, float* __restrict__ part) {
    constexpr int TY = 256 / TX;
    __shared__ float4 red[TY][TX];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * 4;
    const int chunk = blockIdx.y, nchunks = gridDim.y, b = blockIdx.z;
    const int p_lo = chunk * rows_per_chunk, p_hi = min(HW, p_lo + rows_per_chunk);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        const size_t base = (size_t)b * HW * C + c;
        for (int p = p_lo + ty; p < p_hi; p += TY) {
            const float4 xv = *reinterpret_cast<const float4*>(x + base + (size_t)p * C);
            if (y) {
                const float4 yv = *reinterpret_cast<const float4*>(y + base + (size_t)p * C);
                s.x = fmaf(xv.x, yv.x, s.x); s.y = fmaf(xv.y, yv.y, s.y); s.z = fmaf(xv.z, yv.z, s.z); s.w = fmaf(xv.w, yv.w, s.w);
            } else {
                s.x += xv.x; s.y += xv.y; s.z += xv.z; s.w += xv.w;
            }
        }
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float4 t = red[0][tx];
        for (int k = 1; k < TY; ++k) { const float4 o = red[k][tx]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
        *reinterpret_cast<float4*>(part + ((size_t)b * nchunks + chunk) * C + c) = t;
    }
}

__global__ __launch_bounds__(256) void image_dot_finish_kernel(const float* __restrict__ part, int nchunks, int C, int total,
                                                               float scale, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;   // b * C + c
    if (i >= total) return;
    const int b = i / C, c = i - b * C;
    float s = 0.f;
    for (int k = 0; k < nchunks; ++k) s += part[((size_t)b * nchunks + k) * C + c];
    out[i] = s * scale;
}

static int image_dot_rows(int B, int HW, int C) {
    const int col_blocks = (C + 255) / 256;
    long long want = 1024 / ((long long)B * col_blocks);
    if (want < 1) want = 1;
    long long rows = (HW + want - 1) / want;
    if (rows < 32) rows = 32;
    return (int)rows;
}

long long image_dot_ws_floats(int B, int HW, int C) {
    const int rows = image_dot_rows(B, HW, C);
    return (long long)B * ((HW + rows - 1) / rows) * C;
}

int image_dot_f32(const float* x, const float* y, int B, int HW, int C, float scale, float* out, float* ws, hipStream_t st) {
    if (!x || !out || !ws || B <= 0 || HW <= 0 || C <= 0) { set_error("image_dot: bad arguments"); return ICAP_ERR_ARG; }
    if (C % 4 || !aligned16(x) || (y && !aligned16(y)) || !aligned16(ws)) { set_error("image_dot: C %% 4 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    const int rows = image_dot_rows(B, HW, C), nchunks = (HW + rows - 1) / rows, c4 = C / 4;
    const int tx = c4 <= 8 ? 8 : (c4 <= 16 ? 16 : (c4 <= 32 ? 32 : 64));
    const dim3 grid((c4 + tx - 1) / tx, nchunks, B);
    switch (tx) {
        case 8: hipLaunchKernelGGL(image_dot_kernel<8>, grid, dim3(256), 0, st, x, y, HW, C, rows, ws); break;
        case 16: hipLaunchKernelGGL(image_dot_kernel<16>, grid, dim3(256), 0, st, x, y, HW, C, rows, ws); break;
        case 32: hipLaunchKernelGGL(image_dot_kernel<32>, grid, dim3(256), 0, st, x, y, HW, C, rows, ws); break;
        default: hipLaunchKernelGGL(image_dot_kernel<64>, grid, dim3(256), 0, st, x, y, HW, C, rows, ws); break;
    }
    hipLaunchKernelGGL(image_dot_finish_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, ws, nchunks, C, B * C, scale, out);
    return launch_status("image_dot");
}

// ---------------------------------------------------------------- depthwise convolution, input gradient
// dx[b,hi,wi,c] = sum over taps (r,q) with ho*S - pad_t + r == hi, wo*S - pad_l + q == wi of dy[b,ho,wo,c] * w[r*K+q][c]
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              int B, int H, int W, int C4, int K, int S, int pad_t, int pad_l,
                                                              int Ho, int Wo, float* __restrict__ dx) {
    const long long total = (long long)B * H * W * C4;
    const int C = C4 * 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int wi = (int)(t % W); t /= W;
        const int hi = (int)(t % H);
        const int b = (int)(t / H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < K; ++r) {
            const int hn = hi + pad_t - r;
            if (hn < 0 || hn % S) continue;
            const int ho = hn / S;
            if (ho >= Ho) continue;
            for (int q = 0; q < K; ++q) {
                const int wn = wi + pad_l - q;
                if (wn < 0 || wn % S) continue;
                const int wo = wn / S;
                if (wo >= Wo) continue;
                const float4 d = *reinterpret_cast<const float4*>(dy + (((size_t)b * Ho + ho) * Wo + wo) * C + c4 * 4);
                const float4 ww = *reinterpret_cast<const float4*>(w + (size_t)(r * K + q) * C + c4 * 4);
                acc.x = fmaf(d.x, ww.x, acc.x); acc.y = fmaf(d.y, ww.y, acc.y);
                acc.z = fmaf(d.z, ww.z, acc.z); acc.w = fmaf(d.w, ww.w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(dx + i * 4) = acc;
    }
}

int dwconv_bwd_data_f32(const float* dy, const float* w, int B, int H, int W, int C, int K, int S, int pad_t, int pad_l, int Ho,
                        int Wo, float* dx, hipStream_t st) {
    if (!dy || !w || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || S <= 0 || Ho <= 0 || Wo <= 0) {
        set_error("dwconv_bwd_data: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (C % 4 || !aligned16(dy) || !aligned16(w) || !aligned16(dx)) { set_error("dwconv_bwd_data: C %% 4 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    const long long total = (long long)B * H * W * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, w, B, H, W, C / 4, K, S, pad_t, pad_l, Ho, Wo, dx);
    return launch_status("dwconv_bwd_data");
}

// ---------------------------------------------------------------- depthwise convolution, weight gradient
// dw[r*K+q][c] = sum over (b,ho,wo) of dy[b,ho,wo,c] * x[b, ho*S - pad_t + r, wo*S - pad_l + q, c].
// TX threads span 4*TX adjacent channels (one float4 each, K*K float4 accumulators), 256/TX threads share a chunk
// of output pixels; partials [chunk][K*K][C] are added in order by dwconv_wgrad_finish_kernel.
template <int KK, int TX>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x, int B,
                                                                int H, int W, int C, int K, int S, int pad_t, int pad_l, int Ho,
                                                                int Wo, int rows_per_chunk, float* __restrict__ part) {
    constexpr int TY = 256 / TX;
    __shared__ float4 red[TY][TX];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * 4;
    const int chunk = blockIdx.y;
    const long long M = (long long)B * Ho * Wo;
    const long long lo = (long long)chunk * rows_per_chunk, hi = min(M, lo + rows_per_chunk);
    float4 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        for (long long m = lo + ty; m < hi; m += TY) {
            const int wo = (int)(m % Wo);
            const long long t2 = m / Wo;
            const int ho = (int)(t2 % Ho);
            const int b = (int)(t2 / Ho);
            const float4 d = *reinterpret_cast<const float4*>(dy + (size_t)m * C + c);
#pragma unroll
            for (int t = 0; t < KK; ++t) {
                const int r = t / K, q = t - r * K;
                const int hi2 = ho * S - pad_t + r, wi2 = wo * S - pad_l + q;
                if ((unsigned)hi2 < (unsigned)H && (unsigned)wi2 < (unsigned)W) {
                    const float4 xv = *reinterpret_cast<const float4*>(x + (((size_t)b * H + hi2) * W + wi2) * C + c);
                    acc[t].x = fmaf(d.x, xv.x, acc[t].x); acc[t].y = fmaf(d.y, xv.y, acc[t].y);
                    acc[t].z = fmaf(d.z, xv.z, acc[t].z); acc[t].w = fmaf(d.w, xv.w, acc[t].w);
                }
            }
        }
    }
    for (int t = 0; t < KK; ++t) {
        red[ty][tx] = acc[t];
        __syncthreads();
        if (ty == 0 && c < C) {
            float4 s = red[0][tx];
            for (int k = 1; k < TY; ++k) { const float4 o = red[k][tx]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
            *reinterpret_cast<float4*>(part + ((size_t)chunk * KK + t) * C + c) = s;
        }
        __syncthreads();
    }
}

// 64 outputs per workgroup, four lanes per output each adding every fourth chunk (fixed order)
__global__ __launch_bounds__(256) void dwconv_wgrad_finish_kernel(const float* __restrict__ part, int nchunks, int n,
                                                                  float* __restrict__ out) {
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ky = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (i < n)
        for (int k = ky; k < nchunks; k += 4) s += part[(size_t)k * n + i];
    red[ky][cx] = s;
    __syncthreads();
    if (ky == 0 && i < n) out[i] = ((red[0][cx] + red[1][cx]) + red[2][cx]) + red[3][cx];
}

static int dww_rows(long long M, int C) {
    const int col_blocks = (C + 255) / 256;
    long long want = 512 / col_blocks;
    if (want < 1) want = 1;
    long long rows = (M + want - 1) / want;
    if (rows < 32) rows = 32;
    return (int)rows;
}

long long dwconv_bwd_weight_ws_floats(int B, int Ho, int Wo, int C, int K) {
    const long long M = (long long)B * Ho * Wo;
    const int rows = dww_rows(M, C);
    return ((M + rows - 1) / rows) * (long long)K * K * C;
}

template <int KK>
static void dww_launch(int tx, dim3 grid, hipStream_t st, const float* dy, const float* x, int B, int H, int W, int C, int K, int S,
                       int pad_t, int pad_l, int Ho, int Wo, int rows, float* ws) {
    switch (tx) {
        case 8: hipLaunchKernelGGL((dwconv_bwd_weight_kernel<KK, 8>), grid, dim3(256), 0, st, dy, x, B, H, W, C, K, S, pad_t, pad_l, Ho, Wo, rows, ws); break;
        case 16: hipLaunchKernelGGL((dwconv_bwd_weight_kernel<KK, 16>), grid, dim3(256), 0, st, dy, x, B, H, W, C, K, S, pad_t, pad_l, Ho, Wo, rows, ws); break;
        case 32: hipLaunchKernelGGL((dwconv_bwd_weight_kernel<KK, 32>), grid, dim3(256), 0, st, dy, x, B, H, W, C, K, S, pad_t, pad_l, Ho, Wo, rows, ws); break;
        default: hipLaunchKernelGGL((dwconv_bwd_weight_kernel<KK, 64>), grid, dim3(256), 0, st, dy, x, B, H, W, C, K, S, pad_t, pad_l, Ho, Wo, rows, ws); break;
    }
}

int dwconv_bwd_weight_f32(const float* dy, const float* x, int B, int H, int W, int C, int K, int S, int pad_t, int pad_l, int Ho,
                          int Wo, float* dw, float* ws, hipStream_t st) {
    if (!dy || !x || !dw || !ws || B <= 0 || H <= 0 || W <= 0 || C <= 0 || S <= 0 || Ho <= 0 || Wo <= 0) {
        set_error("dwconv_bwd_weight: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (K != 3 && K != 5) { set_error("dwconv_bwd_weight: kernel size %d not supported (3 or 5)", K); return ICAP_ERR_ARG; }
    if (C % 4 || !aligned16(dy) || !aligned16(x) || !aligned16(ws)) { set_error("dwconv_bwd_weight: C %% 4 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    const long long M = (long long)B * Ho * Wo;
    const int rows = dww_rows(M, C);
    const int nchunks = (int)((M + rows - 1) / rows);
    const int c4 = C / 4;
    const int tx = c4 <= 8 ? 8 : (c4 <= 16 ? 16 : (c4 <= 32 ? 32 : 64));
    const dim3 grid((c4 + tx - 1) / tx, nchunks);
    if (K == 3) dww_launch<9>(tx, grid, st, dy, x, B, H, W, C, K, S, pad_t, pad_l, Ho, Wo, rows, ws);
    else dww_launch<25>(tx, grid, st, dy, x, B, H, W, C, K, S, pad_t, pad_l, Ho, Wo, rows, ws);
    const int n = K * K * C;
    hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3((n + 63) / 64), dim3(256), 0, st, ws, nchunks, n, dw);
    return launch_status("dwconv_bwd_weight");
}

// ---------------------------------------------------------------- im2col (stem convolution weight gradient)
// col[m][(r*KW + q)*C + c] = x[b, ho*S - pad_t + r, wo*S - pad_l + q, c] (0 outside), m = (b*Ho + ho)*Wo + wo.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, int B, int H, int W, int C, int KH, int KW, int S,
                                                     int pad_t, int pad_l, int Ho, int Wo, float* __restrict__ col) {
    const int Kc = KH * KW * C;
    const long long total = (long long)B * Ho * Wo * Kc;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % Kc);
        const long long m = i / Kc;
        const int c = k % C, tap = k / C;
        const int r = tap / KW, q = tap - r * KW;
        const int wo = (int)(m % Wo);
        const long long t = m / Wo;
        const int ho = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const int hi = ho * S - pad_t + r, wi = wo * S - pad_l + q;
        col[i] = ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) ? x[(((size_t)b * H + hi) * W + wi) * C + c] : 0.f;
    }
}

int im2col_f32(const float* x, int B, int H, int W, int C, int KH, int KW, int S, int pad_t, int pad_l, int Ho, int Wo, float* col,
               hipStream_t st) {
    if (!x || !col || B <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || S <= 0 || Ho <= 0 || Wo <= 0) {
        set_error("im2col: bad arguments");
        return ICAP_ERR_ARG;
    }
    const long long total = (long long)B * Ho * Wo * KH * KW * C;
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, B, H, W, C, KH, KW, S, pad_t, pad_l, Ho, Wo, col);
    return launch_status("im2col");
}

}  // namespace icap
