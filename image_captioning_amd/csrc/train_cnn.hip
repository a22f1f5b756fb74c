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

// y[b,p,c] = act(x[b,p,c] * s[c] + bi[c]) * g[b][c]: BatchNorm apply + activation + squeeze-excite gate in one pass, from the
// BatchNorm's INPUT - the ungated activation is never written (the gate needs the pooled activation of the whole image
// first, which act_pool_kernel takes without storing anything).  A thread keeps one channel quad (the grid's stride is a
// multiple of the quads per row), so s and bi are loaded once.
template <int ACT>
__global__ __launch_bounds__(256) void act_gate_kernel(const float4* __restrict__ x, const float* __restrict__ s,
                                                       const float* __restrict__ bi, int act, const float4* __restrict__ g, int HW,
                                                       int C4, long long rows, float4* __restrict__ y) {
    const unsigned gtid = blockIdx.x * 256u + threadIdx.x;
    const unsigned stride = gridDim.x * 256u;                 // a multiple of C4
    const unsigned row_step = stride / (unsigned)C4;
    unsigned r = gtid / (unsigned)C4;
    const unsigned c4 = gtid - r * (unsigned)C4;
    const float4 sv = *reinterpret_cast<const float4*>(s + (c4 << 2)), bv = *reinterpret_cast<const float4*>(bi + (c4 << 2));
    const int a_ = ACT >= 0 ? ACT : act;
    constexpr int RU = 4;
    for (; r < (unsigned)rows; r += row_step * RU) {
        float4 xv[RU], gv[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const unsigned rr = min(r + u * row_step, (unsigned)rows - 1u);
            xv[u] = x[(size_t)rr * C4 + c4];
            gv[u] = g[(size_t)(rr / (unsigned)HW) * C4 + c4];
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const unsigned rr = r + u * row_step;
            if (rr >= (unsigned)rows) break;
            y[(size_t)rr * C4 + c4] = make_float4(act_apply(fmaf(xv[u].x, sv.x, bv.x), a_) * gv[u].x, act_apply(fmaf(xv[u].y, sv.y, bv.y), a_) * gv[u].y,
                                                  act_apply(fmaf(xv[u].z, sv.z, bv.z), a_) * gv[u].z, act_apply(fmaf(xv[u].w, sv.w, bv.w), a_) * gv[u].w);
        }
    }
}

int act_gate_f32(const float* x, const float* s, const float* b, int act, const float* g, int B, int HW, int C, float* y, hipStream_t st) {
    if (!x || !s || !b || !g || !y || B <= 0 || HW <= 0 || C <= 0) { set_error("act_gate: bad arguments"); return ICAP_ERR_ARG; }
    const long long rows = (long long)B * HW;
    if (C % 4 || rows * (C / 4) >= (1LL << 31) || !aligned16(x) || !aligned16(y) || !aligned16(s) || !aligned16(b) || !aligned16(g)) {
        set_error("act_gate: C %% 4 == 0, fewer than 2^31 channel quads and 16-byte aligned operands required");
        return ICAP_ERR_ALIGN;
    }
    const int C4 = C / 4;
    int gcd = C4, t256 = 256;
    while (t256) { const int t = gcd % t256; gcd = t256; t256 = t; }
    const long long mult = C4 / gcd;
    long long vb = (rows * C4 + 256LL * 16 - 1) / (256LL * 16);
    if (vb > 16384) vb = 16384;
    vb = (vb + mult - 1) / mult * mult;
    if (act == ACT_SWISH)
        hipLaunchKernelGGL(act_gate_kernel<ACT_SWISH>, dim3((unsigned)vb), dim3(256), 0, st, reinterpret_cast<const float4*>(x), s, b, act,
                           reinterpret_cast<const float4*>(g), HW, C4, rows, reinterpret_cast<float4*>(y));
    else
        hipLaunchKernelGGL(act_gate_kernel<-1>, dim3((unsigned)vb), dim3(256), 0, st, reinterpret_cast<const float4*>(x), s, b, act,
                           reinterpret_cast<const float4*>(g), HW, C4, rows, reinterpret_cast<float4*>(y));
    return launch_status("act_gate");
}

// out[b][c] = scale * sum_p x[b,p,c] * y[b,p,c]  (y == null: sum of x).  TX threads span 4*TX adjacent channels
// (one float4 each), 256/TX threads share the rows of one chunk of one image; chunk partials go to
// part[b][chunk][C] and image_dot_finish_kernel adds them in order.
// With s2 != null the second operand is act(y * s2[c] + b2[c]) - an activation recomputed from its BatchNorm input instead
// of read from a stored copy (ACT2 fixes the activation at compile time).
template <int TX, int ACT2 = ACT_NONE>
__global__ __launch_bounds__(256) void image_dot_kernel(const float* __restrict__ x, const float* __restrict__ y, int HW, int C,
                                                        int rows_per_chunk, float* __restrict__ part,
                                                        const float* __restrict__ s2 = nullptr, const float* __restrict__ b2 = nullptr) {
    constexpr int TY = 256 / TX;
    __shared__ float4 red[TY][TX];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * 4;
    const int chunk = blockIdx.y, nchunks = gridDim.y, b = blockIdx.z;
    const int p_lo = chunk * rows_per_chunk, p_hi = min(HW, p_lo + rows_per_chunk);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        const size_t base = (size_t)b * HW * C + c;
        float4 sv = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s2) { sv = *reinterpret_cast<const float4*>(s2 + c); bv = *reinterpret_cast<const float4*>(b2 + c); }
        // four row positions per iteration, their loads issued before the first is used (rows beyond the chunk read a
        // clamped address and are skipped)
        constexpr int RU = 4;
        for (int p = p_lo + ty; p < p_hi; p += RU * TY) {
            float4 xq[RU], yq[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const size_t o = base + (size_t)min(p + u * TY, p_hi - 1) * C;
                xq[u] = *reinterpret_cast<const float4*>(x + o);
                if (y) yq[u] = *reinterpret_cast<const float4*>(y + o);
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if (p + u * TY >= p_hi) break;
                const float4 xv = xq[u];
                if (y) {
                    float4 yv = yq[u];
                    if (s2) yv = make_float4(act_c<ACT2>(fmaf(yv.x, sv.x, bv.x)), act_c<ACT2>(fmaf(yv.y, sv.y, bv.y)),
                                             act_c<ACT2>(fmaf(yv.z, sv.z, bv.z)), act_c<ACT2>(fmaf(yv.w, sv.w, bv.w)));
                    s.x = fmaf(xv.x, yv.x, s.x); s.y = fmaf(xv.y, yv.y, s.y); s.z = fmaf(xv.z, yv.z, s.z); s.w = fmaf(xv.w, yv.w, s.w);
                } else {
                    s.x += xv.x; s.y += xv.y; s.z += xv.z; s.w += xv.w;
                }
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
    // sixteen chunk partials in flight at a time (one at a time the loop is a chain of nchunks memory round trips), added in
    // chunk order
    float s = 0.f;
    const float* src = part + (size_t)b * nchunks * C + c;
    for (int k0 = 0; k0 < nchunks; k0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(size_t)min(k0 + u, nchunks - 1) * C];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += k0 + u < nchunks ? v[u] : 0.f;
    }
    out[i] = s * scale;
}

static int image_dot_rows(int B, int HW, int C) {
    const int col_blocks = (C + 255) / 256;
    long long want = 2048 / ((long long)B * col_blocks);
    if (want < 1) want = 1;
    long long rows = (HW + want - 1) / want;
    if (rows < 32) rows = 32;
    return (int)rows;
}

long long image_dot_ws_floats(int B, int HW, int C) {
    const int rows = image_dot_rows(B, HW, C);
    return (long long)B * ((HW + rows - 1) / rows) * C;
}

template <int ACT2>
static void image_dot_launch(int tx, dim3 grid, hipStream_t st, const float* x, const float* y, int HW, int C, int rows, float* ws,
                             const float* s2, const float* b2) {
    switch (tx) {
        case 8: hipLaunchKernelGGL((image_dot_kernel<8, ACT2>), grid, dim3(256), 0, st, x, y, HW, C, rows, ws, s2, b2); break;
        case 16: hipLaunchKernelGGL((image_dot_kernel<16, ACT2>), grid, dim3(256), 0, st, x, y, HW, C, rows, ws, s2, b2); break;
        case 32: hipLaunchKernelGGL((image_dot_kernel<32, ACT2>), grid, dim3(256), 0, st, x, y, HW, C, rows, ws, s2, b2); break;
        default: hipLaunchKernelGGL((image_dot_kernel<64, ACT2>), grid, dim3(256), 0, st, x, y, HW, C, rows, ws, s2, b2); break;
    }
}

// s2, b2 (both or neither): the second operand is act(y * s2[c] + b2[c]); act must be ACT_NONE or ACT_SWISH then
int image_dot_act_f32(const float* x, const float* y, const float* s2, const float* b2, int act, int B, int HW, int C, float scale,
                      float* out, float* ws, hipStream_t st) {
    if (!x || !out || !ws || B <= 0 || HW <= 0 || C <= 0 || (!s2) != (!b2) || (s2 && !y)) { set_error("image_dot: bad arguments"); return ICAP_ERR_ARG; }
    if (s2 && act != ACT_NONE && act != ACT_SWISH) { set_error("image_dot: activation %d on the second operand is not built", act); return ICAP_ERR_ARG; }
    if (C % 4 || !aligned16(x) || (y && !aligned16(y)) || !aligned16(ws) || (s2 && (!aligned16(s2) || !aligned16(b2)))) {
        set_error("image_dot: C %% 4 and 16-byte alignment required");
        return ICAP_ERR_ALIGN;
    }
    const int rows = image_dot_rows(B, HW, C), nchunks = (HW + rows - 1) / rows, c4 = C / 4;
    const int tx = c4 <= 8 ? 8 : (c4 <= 16 ? 16 : (c4 <= 32 ? 32 : 64));
    const dim3 grid((c4 + tx - 1) / tx, nchunks, B);
    if (s2 && act == ACT_SWISH) image_dot_launch<ACT_SWISH>(tx, grid, st, x, y, HW, C, rows, ws, s2, b2);
    else image_dot_launch<ACT_NONE>(tx, grid, st, x, y, HW, C, rows, ws, s2, b2);
    hipLaunchKernelGGL(image_dot_finish_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, ws, nchunks, C, B * C, scale, out);
    return launch_status("image_dot");
}

int image_dot_f32(const float* x, const float* y, int B, int HW, int C, float scale, float* out, float* ws, hipStream_t st) {
    return image_dot_act_f32(x, y, nullptr, nullptr, ACT_NONE, B, HW, C, scale, out, ws, st);
}

// y[b,p,c] = act(x[b,p,c] * s[c] + bi[c]) and out[b][c] = scale * sum_p y[b,p,c] in the same pass: the BatchNorm apply in
// front of a squeeze-excite block and the block's global pooling.  y == null: only the pooled sums (act_gate_kernel writes
// the gated activation from x afterwards).  Same thread layout and chunk partials as
// image_dot_kernel; four row positions in flight per thread.
template <int TX, int ACT>      // ACT >= 0: the activation fixed at compile time; < 0: the `act` argument
__global__ __launch_bounds__(256) void act_pool_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                       const float* __restrict__ bi, int act, int HW, int C, int rows_per_chunk,
                                                       float* __restrict__ y, float* __restrict__ part) {
    constexpr int TY = 256 / TX, RU = 4;
    __shared__ float4 red[TY][TX];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * 4;
    const int chunk = blockIdx.y, nchunks = gridDim.y, b = blockIdx.z;
    const int p_lo = chunk * rows_per_chunk, p_hi = min(HW, p_lo + rows_per_chunk);
    const int a_ = ACT >= 0 ? ACT : act;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        const float4 sv = *reinterpret_cast<const float4*>(s + c), bv = *reinterpret_cast<const float4*>(bi + c);
        const size_t base = (size_t)b * HW * C + c;
        for (int p = p_lo + ty; p < p_hi; p += RU * TY) {
            float4 xv[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) xv[u] = *reinterpret_cast<const float4*>(x + base + (size_t)min(p + u * TY, p_hi - 1) * C);
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                if (p + u * TY >= p_hi) break;
                const float4 v = make_float4(act_apply(fmaf(xv[u].x, sv.x, bv.x), a_), act_apply(fmaf(xv[u].y, sv.y, bv.y), a_),
                                             act_apply(fmaf(xv[u].z, sv.z, bv.z), a_), act_apply(fmaf(xv[u].w, sv.w, bv.w), a_));
                if (y) *reinterpret_cast<float4*>(y + base + (size_t)(p + u * TY) * C) = v;
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
        }
    }
    red[ty][tx] = sum;
    __syncthreads();
    if (ty == 0 && c < C) {
        float4 t = red[0][tx];
        for (int k = 1; k < TY; ++k) { const float4 o = red[k][tx]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
        *reinterpret_cast<float4*>(part + ((size_t)b * nchunks + chunk) * C + c) = t;
    }
}

template <int ACT>
static void act_pool_launch(int tx, dim3 grid, hipStream_t st, const float* x, const float* s, const float* b, int act, int HW, int C,
                            int rows, float* y, float* ws) {
    switch (tx) {
        case 8: hipLaunchKernelGGL((act_pool_kernel<8, ACT>), grid, dim3(256), 0, st, x, s, b, act, HW, C, rows, y, ws); break;
        case 16: hipLaunchKernelGGL((act_pool_kernel<16, ACT>), grid, dim3(256), 0, st, x, s, b, act, HW, C, rows, y, ws); break;
        case 32: hipLaunchKernelGGL((act_pool_kernel<32, ACT>), grid, dim3(256), 0, st, x, s, b, act, HW, C, rows, y, ws); break;
        default: hipLaunchKernelGGL((act_pool_kernel<64, ACT>), grid, dim3(256), 0, st, x, s, b, act, HW, C, rows, y, ws); break;
    }
}

// channel-quad lanes per workgroup: the width among 8 / 16 / 32 / 64 with the fewest idle lanes in the last column block
static int pool_tx(int c4) {
    int best = 64, best_waste = 1 << 30;
    for (int tx = 64; tx >= 8; tx >>= 1) {
        const int waste = (c4 + tx - 1) / tx * tx - c4;
        if (waste < best_waste) { best_waste = waste; best = tx; }
    }
    return best;
}

// rows of one image per workgroup: about 2048 workgroups in all, at least 32 rows each
static int act_pool_rows(int B, int HW, int C) {
    const int c4 = C / 4, tx = pool_tx(c4), col_blocks = (c4 + tx - 1) / tx;
    long long want = 2048 / ((long long)B * col_blocks);
    if (want < 1) want = 1;
    long long rows = (HW + want - 1) / want;
    if (rows < 32) rows = 32;
    return (int)rows;
}

long long act_pool_ws_floats(int B, int HW, int C) {
    const int rows = act_pool_rows(B, HW, C);
    return (long long)B * ((HW + rows - 1) / rows) * C;
}

int act_pool_f32(const float* x, const float* s, const float* b, int act, int B, int HW, int C, float scale, float* y, float* out,
                 float* ws, hipStream_t st) {
    if (!x || !s || !b || !out || !ws || B <= 0 || HW <= 0 || C <= 0) { set_error("act_pool: bad arguments"); return ICAP_ERR_ARG; }
    if (C % 4 || !aligned16(x) || (y && !aligned16(y)) || !aligned16(s) || !aligned16(b) || !aligned16(ws)) {
        set_error("act_pool: C %% 4 and 16-byte alignment required");
        return ICAP_ERR_ALIGN;
    }
    const int rows = act_pool_rows(B, HW, C), nchunks = (HW + rows - 1) / rows, c4 = C / 4, tx = pool_tx(c4);
    const dim3 grid((c4 + tx - 1) / tx, nchunks, B);
    if (act == ACT_SWISH) act_pool_launch<ACT_SWISH>(tx, grid, st, x, s, b, act, HW, C, rows, y, ws);
    else act_pool_launch<-1>(tx, grid, st, x, s, b, act, HW, C, rows, y, ws);
    hipLaunchKernelGGL(image_dot_finish_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, ws, nchunks, C, B * C, scale, out);
    return launch_status("act_pool");
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

// Stride-2 form (the four down-sampling blocks of EfficientNet-B0: 3x3 with TF-'same' padding 0 / 1, 5x5 with 1 / 2).  Input
// pixel (hi, wi) receives tap (r, q) of output pixel ((hi + PAD - r) / 2, (wi + PAD - q) / 2) when both differences are even, so
// the four pixels of a 2 x 2 input block (a, b) use the same NDY x NDY window of dy around (a, b) with disjoint tap sets.  A
// thread owns one channel quad and TB = 4 adjacent blocks of one block row: it loads the NDY x (TB + NDY - 1) window once
// (every load unconditional from a clamped address, zeroed by a select) and produces 16 input pixels - 1.3 (3x3) or 2.3 (5x5)
// dy loads per pixel quad instead of up to 4 / 9 conditional ones, and no division or parity test per tap.
template <int K, int PAD>
__global__ __launch_bounds__(256) void dwconv_bwd_data_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w, int B, int H,
                                                                 int W, int C4, int Ho, int Wo, int nstrips_w, int nstrips,
                                                                 float* __restrict__ dx) {
    constexpr int TB = 4;
    constexpr int DMIN = -((K - 1 - PAD) / 2);                   // smallest (i + PAD - r) / 2: i = 0 (or 1), r = K - 1 (or K - 2)
    constexpr int DMAX = (1 + PAD) / 2;                           // largest: i = 1, r = (1 + PAD) % 2
    constexpr int NDY = DMAX - DMIN + 1, NW = TB + NDY - 1;
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)B * nstrips * C4;
    if (gi >= total) return;
    const int cq = (int)(gi % C4);
    const long long t1 = gi / C4;
    const int strip = (int)(t1 % nstrips), b = (int)(t1 / nstrips);
    const int a = strip / nstrips_w, b0 = (strip - a * nstrips_w) * TB;
    const int C = C4 * 4, c = cq * 4;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 d[NDY][NW];
#pragma unroll
    for (int u = 0; u < NDY; ++u) {
        const int ho = a + DMIN + u;
        const bool rok = (unsigned)ho < (unsigned)Ho;
        const float* row = dy + (((size_t)b * Ho + (rok ? ho : 0)) * Wo) * C + c;
#pragma unroll
        for (int v = 0; v < NW; ++v) {
            const int wo = b0 + DMIN + v;
            const bool ok = rok && (unsigned)wo < (unsigned)Wo;
            d[u][v] = *reinterpret_cast<const float4*>(row + (size_t)(ok ? wo : 0) * C);
            if (!ok) d[u][v] = z4;
        }
    }
    float4 acc[TB][2][2];
#pragma unroll
    for (int t = 0; t < TB; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = z4;
#pragma unroll
    for (int r = 0; r < K; ++r) {
        constexpr int dummy = 0; (void)dummy;
        const int i = (r + PAD) & 1;                 // the block row parity this filter row serves
        const int u = (i + PAD - r) / 2 - DMIN;      // (i + PAD - r) is even
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const int j = (q + PAD) & 1;
            const int v0 = (j + PAD - q) / 2 - DMIN;
            const float4 ww = *reinterpret_cast<const float4*>(w + (size_t)(r * K + q) * C + c);
#pragma unroll
            for (int t = 0; t < TB; ++t) {
                const float4 dv = d[u][v0 + t];
                acc[t][i][j].x = fmaf(dv.x, ww.x, acc[t][i][j].x); acc[t][i][j].y = fmaf(dv.y, ww.y, acc[t][i][j].y);
                acc[t][i][j].z = fmaf(dv.z, ww.z, acc[t][i][j].z); acc[t][i][j].w = fmaf(dv.w, ww.w, acc[t][i][j].w);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int hi = 2 * a + i;
        if (hi >= H) continue;
        float* orow = dx + (((size_t)b * H + hi) * W) * C + c;
#pragma unroll
        for (int t = 0; t < TB; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int wi = 2 * (b0 + t) + j;
                if (wi < W) *reinterpret_cast<float4*>(orow + (size_t)wi * C) = acc[t][i][j];
            }
    }
}

int dwconv_bwd_data_f32(const float* dy, const float* w, int B, int H, int W, int C, int K, int S, int pad_t, int pad_l, int Ho,
                        int Wo, float* dx, hipStream_t st) {
    if (!dy || !w || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || S <= 0 || Ho <= 0 || Wo <= 0) {
        set_error("dwconv_bwd_data: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (C % 4 || !aligned16(dy) || !aligned16(w) || !aligned16(dx)) { set_error("dwconv_bwd_data: C %% 4 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    if (S == 2 && pad_t == pad_l && ((K == 3 && pad_t == 0) || (K == 5 && pad_t == 1)) && Ho == (H + 1) / 2 && Wo == (W + 1) / 2) {
        // every input pixel lies in a 2 x 2 block whose window of dy is inside [0, Ho) x [0, Wo) or zero
        const int nsw = ((W + 1) / 2 + 3) / 4, nstrips = ((H + 1) / 2) * nsw;
        const long long threads = (long long)B * nstrips * (C / 4);
        const long long nb = (threads + 255) / 256;
        if (nb > 0x7fffffffLL) { set_error("dwconv_bwd_data: problem too large"); return ICAP_ERR_LIMIT; }
        if (K == 3) hipLaunchKernelGGL((dwconv_bwd_data_s2_kernel<3, 0>), dim3((unsigned)nb), dim3(256), 0, st, dy, w, B, H, W, C / 4, Ho, Wo, nsw, nstrips, dx);
        else hipLaunchKernelGGL((dwconv_bwd_data_s2_kernel<5, 1>), dim3((unsigned)nb), dim3(256), 0, st, dy, w, B, H, W, C / 4, Ho, Wo, nsw, nstrips, dx);
        return launch_status("dwconv_bwd_data");
    }
    const long long total = (long long)B * H * W * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, w, B, H, W, C / 4, K, S, pad_t, pad_l, Ho, Wo, dx);
    return launch_status("dwconv_bwd_data");
}

// ---------------------------------------------------------------- depthwise convolution, weight gradient
// dw[r*K+q][c] = sum over (b,ho,wo) of dy[b,ho,wo,c] * x[b, ho*S - pad_t + r, wo*S - pad_l + q, c].
// A workgroup owns one filter row r (grid.z); a thread owns a channel quad (K float4 accumulators) and walks strips of SW
// adjacent output pixels of one output row: the strip's dy values and the input row segment under the strip are loaded
// once and feed all K taps of all SW outputs ((SW-1)*S+K + SW loads instead of SW*(K+1)).  Every load is unconditional
// from a clamped address (zeroed by a select), so a strip's loads are all in flight before the first is consumed, and
// the small register footprint (~80) keeps enough waves resident to hide their latency.
// TX threads span 4*TX adjacent channels, 256/TX threads share a chunk of strips; partials [chunk][K*K][C] are added in
// order by dwconv_wgrad_finish_kernel.
template <int K, int S, int SW, int TX>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x, int B,
                                                                int H, int W, int C, int pad_t, int pad_l, int Ho, int Wo,
                                                                int strips_w, int nstrips, int strips_per_chunk,
                                                                float* __restrict__ part, int col_blocks, int npairs) {
    constexpr int KK = K * K, TY = 256 / TX, NIN = (SW - 1) * S + K;
    __shared__ float4 red[TY][TX];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    // The K workgroups (filter rows) of one (channel block, chunk) pair read the same dy and neighbouring x rows: they take
    // consecutive slots of ONE XCD (workgroup ids go to the XCDs round-robin), so that XCD's L2 serves K - 1 of the K reads.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pair = (slot / K) * 8 + xcd;
    if (pair >= npairs) return;
    const int r = slot % K;                // filter row of this workgroup: K accumulators per thread instead of K*K
    const int chunk = pair / col_blocks;
    const int c = ((pair - chunk * col_blocks) * TX + tx) * 4;
    const int lo = chunk * strips_per_chunk, hi = min(nstrips, lo + strips_per_chunk);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q) acc[q] = z4;
    if (c < C) {
#pragma unroll 2
        for (int sidx = lo + ty; sidx < hi; sidx += TY) {
            const int sw = sidx % strips_w;
            const int t2 = sidx / strips_w;
            const int ho = t2 % Ho;
            const int b = t2 / Ho;
            const int wo0 = sw * SW;
            const int hh = ho * S - pad_t + r, wi0 = wo0 * S - pad_l;
            if ((unsigned)hh >= (unsigned)H) continue;       // this filter row falls into the padding (image borders only)
            const float* dyr = dy + (((size_t)b * Ho + ho) * Wo + wo0) * C + c;
            const float* xr = x + ((size_t)b * H + hh) * W * C + c;
            float4 d[SW], v[NIN];
#pragma unroll
            for (int o = 0; o < SW; ++o) {
                const bool ok = wo0 + o < Wo;
                d[o] = *reinterpret_cast<const float4*>(ok ? dyr + (size_t)o * C : dyr);      // always a valid address
                if (!ok) d[o] = z4;
            }
#pragma unroll
            for (int j = 0; j < NIN; ++j) {
                const int wi = wi0 + j;
                const bool ok = (unsigned)wi < (unsigned)W;
                v[j] = *reinterpret_cast<const float4*>(ok ? xr + (size_t)wi * C : xr);
                if (!ok) v[j] = z4;
            }
#pragma unroll
            for (int q = 0; q < K; ++q)
#pragma unroll
                for (int o = 0; o < SW; ++o) {
                    const float4 xv = v[o * S + q];
                    acc[q].x = fmaf(d[o].x, xv.x, acc[q].x); acc[q].y = fmaf(d[o].y, xv.y, acc[q].y);
                    acc[q].z = fmaf(d[o].z, xv.z, acc[q].z); acc[q].w = fmaf(d[o].w, xv.w, acc[q].w);
                }
        }
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
        red[ty][tx] = acc[q];
        __syncthreads();
        if (ty == 0 && c < C) {
            float4 s = red[0][tx];
#pragma unroll 1
            for (int k = 1; k < TY; ++k) { const float4 o = red[k][tx]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
            *reinterpret_cast<float4*>(part + ((size_t)chunk * KK + r * K + q) * C + c) = s;
        }
        __syncthreads();
    }
}

// 8 outputs per workgroup, 32 lanes per output each adding every 32nd chunk with sixteen loads in flight (fixed order)
__global__ __launch_bounds__(256) void dwconv_wgrad_finish_kernel(const float* __restrict__ part, int nchunks, int n,
                                                                  float* __restrict__ out) {
    __shared__ float red[32][8];
    const int cx = threadIdx.x & 7, ky = threadIdx.x >> 3;
    const int i = blockIdx.x * 8 + cx;
    float s = 0.f;
    if (i < n) {
        for (int k0 = ky; k0 < nchunks; k0 += 32 * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + 32 * u;
                v[u] = k < nchunks ? part[(size_t)k * n + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
    }
    red[ky][cx] = s;
    __syncthreads();
    if (ky == 0 && i < n) {
        float t = red[0][cx];
#pragma unroll
        for (int k = 1; k < 32; ++k) t += red[k][cx];
        out[i] = t;
    }
}

constexpr int DWW_SW = 4;   // strip width: 4 keeps a filter row's loads + K*K accumulators within ~128 registers

// strips per chunk: about 512 chunks (times K filter rows) in all, at least 8 strips per thread row
static int dww_strips_per_chunk(long long nstrips, int C) {
    const int col_blocks = (C + 255) / 256;
    long long want = 512 / col_blocks;
    if (want < 1) want = 1;
    long long per = (nstrips + want - 1) / want;
    if (per < 8) per = 8;
    return (int)per;
}

static long long dww_nstrips(int B, int Ho, int Wo, int K, int S) {
    const int sw = DWW_SW;
    (void)K; (void)S;
    return (long long)B * Ho * ((Wo + sw - 1) / sw);
}

long long dwconv_bwd_weight_ws_floats(int B, int Ho, int Wo, int C, int K) {
    const long long ns = dww_nstrips(B, Ho, Wo, K, 1);
    const int per = dww_strips_per_chunk(ns, C);
    return ((ns + per - 1) / per) * (long long)K * K * C;
}

template <int K, int S, int SW>
static void dww_launch(int tx, dim3 grid, hipStream_t st, const float* dy, const float* x, int B, int H, int W, int C, int pad_t,
                       int pad_l, int Ho, int Wo, int strips_w, int nstrips, int per, float* ws, int col_blocks, int npairs) {
    switch (tx) {
        case 8: hipLaunchKernelGGL((dwconv_bwd_weight_kernel<K, S, SW, 8>), grid, dim3(256), 0, st, dy, x, B, H, W, C, pad_t, pad_l, Ho, Wo, strips_w, nstrips, per, ws, col_blocks, npairs); break;
        case 16: hipLaunchKernelGGL((dwconv_bwd_weight_kernel<K, S, SW, 16>), grid, dim3(256), 0, st, dy, x, B, H, W, C, pad_t, pad_l, Ho, Wo, strips_w, nstrips, per, ws, col_blocks, npairs); break;
        case 32: hipLaunchKernelGGL((dwconv_bwd_weight_kernel<K, S, SW, 32>), grid, dim3(256), 0, st, dy, x, B, H, W, C, pad_t, pad_l, Ho, Wo, strips_w, nstrips, per, ws, col_blocks, npairs); break;
        default: hipLaunchKernelGGL((dwconv_bwd_weight_kernel<K, S, SW, 64>), grid, dim3(256), 0, st, dy, x, B, H, W, C, pad_t, pad_l, Ho, Wo, strips_w, nstrips, per, ws, col_blocks, npairs); break;
    }
}

int dwconv_bwd_weight_f32(const float* dy, const float* x, int B, int H, int W, int C, int K, int S, int pad_t, int pad_l, int Ho,
                          int Wo, float* dw, float* ws, hipStream_t st) {
    if (!dy || !x || !dw || !ws || B <= 0 || H <= 0 || W <= 0 || C <= 0 || S <= 0 || Ho <= 0 || Wo <= 0) {
        set_error("dwconv_bwd_weight: bad arguments");
        return ICAP_ERR_ARG;
    }
    if ((K != 3 && K != 5) || (S != 1 && S != 2)) { set_error("dwconv_bwd_weight: kernel %d stride %d not supported (3 or 5, stride 1 or 2)", K, S); return ICAP_ERR_ARG; }
    if (C % 4 || !aligned16(dy) || !aligned16(x) || !aligned16(ws)) { set_error("dwconv_bwd_weight: C %% 4 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    const long long ns = dww_nstrips(B, Ho, Wo, K, S);
    if (ns > 0x7fffffffLL) { set_error("dwconv_bwd_weight: problem too large"); return ICAP_ERR_LIMIT; }
    const int nstrips = (int)ns;
    const int per = dww_strips_per_chunk(ns, C);
    const int nchunks = (nstrips + per - 1) / per;
    const int c4 = C / 4;
    const int tx = c4 <= 8 ? 8 : (c4 <= 16 ? 16 : (c4 <= 32 ? 32 : 64));
    const int col_blocks = (c4 + tx - 1) / tx;
    const long long npairs_ll = (long long)col_blocks * nchunks;
    if (((npairs_ll + 7) / 8) * K * 8 > 0x7fffffffLL) { set_error("dwconv_bwd_weight: problem too large"); return ICAP_ERR_LIMIT; }
    const int npairs = (int)npairs_ll;
    const dim3 grid((unsigned)(((npairs + 7) / 8) * K * 8));
    if (K == 3 && S == 1) dww_launch<3, 1, DWW_SW>(tx, grid, st, dy, x, B, H, W, C, pad_t, pad_l, Ho, Wo, (Wo + DWW_SW - 1) / DWW_SW, nstrips, per, ws, col_blocks, npairs);
    else if (K == 3) dww_launch<3, 2, DWW_SW>(tx, grid, st, dy, x, B, H, W, C, pad_t, pad_l, Ho, Wo, (Wo + DWW_SW - 1) / DWW_SW, nstrips, per, ws, col_blocks, npairs);
    else if (S == 1) dww_launch<5, 1, DWW_SW>(tx, grid, st, dy, x, B, H, W, C, pad_t, pad_l, Ho, Wo, (Wo + DWW_SW - 1) / DWW_SW, nstrips, per, ws, col_blocks, npairs);
    else dww_launch<5, 2, DWW_SW>(tx, grid, st, dy, x, B, H, W, C, pad_t, pad_l, Ho, Wo, (Wo + DWW_SW - 1) / DWW_SW, nstrips, per, ws, col_blocks, npairs);
    const int n = K * K * C;
    hipLaunchKernelGGL(dwconv_wgrad_finish_kernel, dim3((n + 7) / 8), dim3(256), 0, st, ws, nchunks, n, dw);
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

// ---------------------------------------------------------------- matrix transpose
// dst[c][r] = src[r][c] through a 32 x 33 LDS tile (reads and writes both run along rows).  The input-gradient GEMM
// dX = dY W wants W with the contraction axis contiguous to run on the implicit-GEMM kernel: weights are at most a few
// MB, so a per-step transposed copy costs microseconds.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int rows, int cols, int lds_, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < rows && c < cols) tile[ty + 8 * i][tx] = src[(size_t)r * lds_ + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < rows && c < cols) dst[(size_t)c * rows + r] = tile[tx][ty + 8 * i];
    }
}

int transpose_f32(const float* src, int rows, int cols, int ld_src, float* dst, hipStream_t st) {
    if (!src || !dst || rows <= 0 || cols <= 0 || ld_src < cols) { set_error("transpose: bad arguments"); return ICAP_ERR_ARG; }
    if ((cols + 31) / 32 > 65535 || (rows + 31) / 32 > 65535) { set_error("transpose: matrix too large"); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, src, rows, cols, ld_src, dst);
    return launch_status("transpose");
}

// ---------------------------------------------------------------- squeeze-excite in training mode
// efficientnet_pytorch's squeeze-excite branch of an MBConv block: pooled -> _se_reduce (1x1 conv = FC, bias) -> swish ->
// _se_expand (FC, bias) -> sigmoid -> gate.  As separate launches it is six tiny kernels forward and eight backward per
// block, each a few microseconds of launch latency on a [B, C] problem; here it is one kernel forward and two backward.
// Weights in the parameters' own layout: wr [S][C] (= _se_reduce.weight), we [C][S] (= _se_expand.weight).
constexpr int SET_MAX_C = 2048, SET_MAX_S = 128;

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one workgroup of SE_NT threads per image: r_pre = wr . pooled + br, e = we . swish(r_pre) + be, gate = sigmoid(e).
// The problem is tiny and every phase is a chain of dependent memory round trips, so the time is set by how many loads one
// image keeps in flight: sixteen waves, each taking its squeeze units two at a time.
constexpr int SE_NT = 1024, SE_NW = SE_NT / 64;
constexpr int SE_CL = 16, SE_NS = 256 / SE_CL;       // se_train_bwd_weights_kernel: channels and image slices per workgroup

__global__ __launch_bounds__(SE_NT) void se_train_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ wr,
                                                             const float* __restrict__ br, const float* __restrict__ we,
                                                             const float* __restrict__ be, float* __restrict__ r_pre,
                                                             float* __restrict__ e, float* __restrict__ gate, int C, int S) {
    __shared__ float pl[SET_MAX_C];
    __shared__ float sq[SET_MAX_S];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int c = tid; c < C; c += SE_NT) pl[c] = pooled[(size_t)b * C + c];
    __syncthreads();
    for (int j = wave; j < S; j += 2 * SE_NW) {
        const int j2 = j + SE_NW < S ? j + SE_NW : j;        // the second unit of this pass (a repeat of the first past the end)
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
        for (int c = lane; c < C; c += 64) {
            const float p = pl[c];
            a0 = fmaf(wr[(size_t)j * C + c], p, a0);
            a1 = fmaf(wr[(size_t)j2 * C + c], p, a1);
        }
        a0 = wave_sum_f(a0) + br[j];
        a1 = wave_sum_f(a1) + br[j2];
        if (lane == 0) {
            r_pre[(size_t)b * S + j] = a0;
            sq[j] = a0 * sigmoid_fast(a0);
            if (j2 != j) {
                r_pre[(size_t)b * S + j2] = a1;
                sq[j2] = a1 * sigmoid_fast(a1);
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += SE_NT) {
        float a = be[c];
        const float* w = we + (size_t)c * S;
#pragma unroll 8
        for (int j = 0; j < S; ++j) a = fmaf(w[j], sq[j], a);
        e[(size_t)b * C + c] = a;
        gate[(size_t)b * C + c] = sigmoid_fast(a);
    }
}

// one workgroup per image: de = dgate * sigmoid'(e); dr_pre = (we^T de) * swish'(r_pre); dpooled = wr^T dr_pre
__global__ __launch_bounds__(SE_NT) void se_train_bwd_image_kernel(const float* __restrict__ dgate, const float* __restrict__ e,
                                                                 const float* __restrict__ r_pre, const float* __restrict__ wr,
                                                                 const float* __restrict__ we, float* __restrict__ de,
                                                                 float* __restrict__ dr_pre, float* __restrict__ dpooled, int C,
                                                                 int S) {
    __shared__ float dl[SET_MAX_C];
    __shared__ float dr[SET_MAX_S];
    __shared__ float part2[SE_NW][SET_MAX_S];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int c = tid; c < C; c += SE_NT) {
        const float g = sigmoid_fast(e[(size_t)b * C + c]);
        const float d = dgate[(size_t)b * C + c] * g * (1.f - g);
        dl[c] = d;
        de[(size_t)b * C + c] = d;
    }
    __syncthreads();
    // we^T de: a lane walks channels lane, lane + 64, ... of its wave's share and reads their rows of `we` contiguously
    // (column-wise, one squeeze unit at a time, every lane touched a different cache line per load); the S partial sums of
    // a wave meet in LDS and the first S threads add the waves' shares in order
    {
        constexpr int JB = 16;           // squeeze units per sweep over the channels
        const int cq = (C + SE_NW - 1) / SE_NW, c_lo = min(C, wave * cq), c_hi = min(C, c_lo + cq);
        for (int j0 = 0; j0 < S; j0 += JB) {
            float acc[JB];
#pragma unroll
            for (int j = 0; j < JB; ++j) acc[j] = 0.f;
            for (int c = c_lo + lane; c < c_hi; c += 64) {
                const float d = dl[c];
                const float* w = we + (size_t)c * S + j0;
#pragma unroll
                for (int j = 0; j < JB; ++j) acc[j] = fmaf(w[min(j, S - 1 - j0)], j0 + j < S ? d : 0.f, acc[j]);
            }
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                const float t = wave_sum_f(acc[j]);
                if (lane == 0 && j0 + j < S) part2[wave][j0 + j] = t;
            }
        }
    }
    __syncthreads();
    if (tid < S) {
        const int j = tid;
        float a = part2[0][j];
#pragma unroll
        for (int k = 1; k < SE_NW; ++k) a += part2[k][j];
        const float x = r_pre[(size_t)b * S + j], sg = sigmoid_fast(x);
        const float d = a * sg * (1.f + x * (1.f - sg));
        dr[j] = d;
        dr_pre[(size_t)b * S + j] = d;
    }
    __syncthreads();
    for (int c = tid; c < C; c += SE_NT) {
        float a = 0.f;
#pragma unroll 8
        for (int j = 0; j < S; ++j) a = fmaf(wr[(size_t)j * C + c], dr[j], a);
        dpooled[(size_t)b * C + c] = a;
    }
}

// parameter gradients: a workgroup owns SE_CL channels x 16 squeeze units; its 256 threads are SE_CL channels x SE_NS image
// slices (image b goes to slice b mod SE_NS), each slice sums its images in order and the slices are added in order.  Few
// channels per workgroup on purpose: the loop over the images is a chain of memory round trips, so the launch wants many
// workgroups (216 at C = 1152, S = 48) rather than wide ones
__global__ __launch_bounds__(256) void se_train_bwd_weights_kernel(const float* __restrict__ de, const float* __restrict__ dr_pre,
                                                                   const float* __restrict__ r_pre, const float* __restrict__ pooled,
                                                                   float* __restrict__ dwe, float* __restrict__ dbe,
                                                                   float* __restrict__ dwr, float* __restrict__ dbr, int B, int C,
                                                                   int S) {
    constexpr int SG = 16, BT = 256, CL = SE_CL, NS = SE_NS;
    __shared__ float rl[BT][SG], dl[BT][SG];
    __shared__ float red[NS - 1][CL][2 * SG + 1];
    const int tid = threadIdx.x, cl = tid % CL, slice = tid / CL;
    const int c = blockIdx.x * CL + cl, s0 = blockIdx.y * SG;
    const bool ok = c < C;
    float awe[SG], awr[SG], sbe = 0.f;
#pragma unroll
    for (int j = 0; j < SG; ++j) { awe[j] = 0.f; awr[j] = 0.f; }
    for (int b0 = 0; b0 < B; b0 += BT) {
        __syncthreads();
        for (int i = tid; i < BT * SG; i += 256) {
            const int bb = i / SG, j = i - bb * SG;
            const bool in = b0 + bb < B && s0 + j < S;
            const float x = in ? r_pre[(size_t)(b0 + bb) * S + s0 + j] : 0.f;
            rl[bb][j] = x * sigmoid_fast(x);
            dl[bb][j] = in ? dr_pre[(size_t)(b0 + bb) * S + s0 + j] : 0.f;
        }
        __syncthreads();
        const int nb = min(BT, B - b0);
        // eight images of this slice per round, their loads issued before the first is used (one image at a time the loop is a
        // chain of memory round trips); images beyond the batch read a clamped address and count as zero
        constexpr int IU = 8;
        const int cc = ok ? c : 0;
        for (int bb = slice; bb < nb; bb += NS * IU) {
            float dv[IU], pvv[IU];
#pragma unroll
            for (int u = 0; u < IU; ++u) {
                const int bi = min(bb + NS * u, nb - 1);
                dv[u] = de[(size_t)(b0 + bi) * C + cc];
                pvv[u] = pooled[(size_t)(b0 + bi) * C + cc];
            }
#pragma unroll
            for (int u = 0; u < IU; ++u) {
                const int bi = bb + NS * u;
                if (bi >= nb) break;
                const float d = ok ? dv[u] : 0.f, pv = ok ? pvv[u] : 0.f;
                sbe += d;
#pragma unroll
                for (int j = 0; j < SG; ++j) {
                    awe[j] = fmaf(d, rl[bi][j], awe[j]);
                    awr[j] = fmaf(dl[bi][j], pv, awr[j]);
                }
            }
        }
    }
    if (slice > 0) {
#pragma unroll
        for (int j = 0; j < SG; ++j) { red[slice - 1][cl][j] = awe[j]; red[slice - 1][cl][SG + j] = awr[j]; }
        red[slice - 1][cl][2 * SG] = sbe;
    }
    __syncthreads();
    if (slice == 0 && ok) {
        for (int k = 0; k < NS - 1; ++k) {
#pragma unroll
            for (int j = 0; j < SG; ++j) { awe[j] += red[k][cl][j]; awr[j] += red[k][cl][SG + j]; }
            sbe += red[k][cl][2 * SG];
        }
#pragma unroll
        for (int j = 0; j < SG; ++j)
            if (s0 + j < S) {
                dwe[(size_t)c * S + s0 + j] = awe[j];
                dwr[(size_t)(s0 + j) * C + c] = awr[j];
            }
        if (blockIdx.y == 0) dbe[c] = sbe;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid < S) {
        float a = 0.f;
        for (int b = 0; b < B; b += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = dr_pre[(size_t)min(b + u, B - 1) * S + tid];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += b + u < B ? v[u] : 0.f;
        }
        dbr[tid] = a;
    }
}

static int se_train_args(const char* what, int B, int C, int S) {
    if (B <= 0 || C <= 0 || S <= 0) { set_error("%s: bad arguments", what); return ICAP_ERR_ARG; }
    if (C > SET_MAX_C || S > SET_MAX_S) { set_error("%s: C=%d S=%d exceed %d/%d", what, C, S, SET_MAX_C, SET_MAX_S); return ICAP_ERR_LIMIT; }
    return ICAP_OK;
}

int se_train_fwd_f32(const float* pooled, const float* wr, const float* br, const float* we, const float* be, float* r_pre,
                     float* e, float* gate, int B, int C, int S, hipStream_t st) {
    if (!pooled || !wr || !br || !we || !be || !r_pre || !e || !gate) { set_error("se_train_fwd: null pointer"); return ICAP_ERR_ARG; }
    if (const int rc = se_train_args("se_train_fwd", B, C, S)) return rc;
    hipLaunchKernelGGL(se_train_fwd_kernel, dim3(B), dim3(SE_NT), 0, st, pooled, wr, br, we, be, r_pre, e, gate, C, S);
    return launch_status("se_train_fwd");
}

int se_train_bwd_f32(const float* dgate, const float* e, const float* r_pre, const float* pooled, const float* wr, const float* we,
                     float* de, float* dr_pre, float* dpooled, float* dwe, float* dbe, float* dwr, float* dbr, int B, int C, int S,
                     hipStream_t st) {
    if (!dgate || !e || !r_pre || !pooled || !wr || !we || !de || !dr_pre || !dpooled || !dwe || !dbe || !dwr || !dbr) {
        set_error("se_train_bwd: null pointer");
        return ICAP_ERR_ARG;
    }
    if (const int rc = se_train_args("se_train_bwd", B, C, S)) return rc;
    hipLaunchKernelGGL(se_train_bwd_image_kernel, dim3(B), dim3(SE_NT), 0, st, dgate, e, r_pre, wr, we, de, dr_pre, dpooled, C, S);
    hipLaunchKernelGGL(se_train_bwd_weights_kernel, dim3((C + SE_CL - 1) / SE_CL, (S + 15) / 16), dim3(256), 0, st, de, dr_pre, r_pre, pooled, dwe,
                       dbe, dwr, dbr, B, C, S);
    return launch_status("se_train_bwd");
}

}  // namespace icap
