// EfficientNet-B0 stem (3x3 stride-2 convolution of the planar fp32 image, folded BN, swish) fused with block 0's depthwise
// 3x3 stride-1 convolution (+ folded BN + swish + the squeeze-excite partial sums), on the vector ALUs alone:
//   * the 33 x 33 x 3 image patch under the tile is fetched into LDS in one go (every load of the tile in flight at once; a
//     wave-load takes two patch rows, so the addressing is a handful of instructions per load);
//   * a thread owns ONE stem pixel of the tile's 16 x 16 stem region (the 14 x 14 output tile + the depthwise halo) and all
//     32 channels: it reads its 27 image values from the patch tap by tap, the weights w[k][c] are the same for every lane, so they come
//     through the scalar cache as instruction operands, and the 864 multiply-adds are packed two channels per instruction;
//     nothing depends on a matrix-instruction chain and ~100 registers leave five waves per SIMD to hide the loads;
//   * BN + swish, zero for stem pixels outside the stem's output (the depthwise convolution pads with zeros), then the tile
//     goes to LDS once ([256 pixels][32 + 4 channels]);
//   * depthwise phase: a thread owns a strip of 7 output pixels x 4 channels, slides over the 3 x 9 window of LDS rows,
//     applies BN + swish, stores 16 bytes per pixel (eight lanes = one pixel's 128-byte line) and adds up its outputs for
//     the per-tile channel sums.
#include <type_traits>
#include "common.hpp"

namespace icap {

typedef float sv_f32x2 __attribute__((ext_vector_type(2)));
typedef float sv_f32x4 __attribute__((ext_vector_type(4)));

constexpr int SV_C = 32, SV_TH = 14, SV_TW = 14, SV_EH = SV_TH + 2, SV_EW = SV_TW + 2, SV_LDE = SV_C + 4, SV_SW = 7;
// swish of two values with packed arithmetic around the four transcendentals
__device__ __forceinline__ sv_f32x2 sv_swish2(sv_f32x2 v) {
    const sv_f32x2 t = v * sv_f32x2{-1.4426950408889634f, -1.4426950408889634f};
    const sv_f32x2 d = sv_f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + sv_f32x2{1.f, 1.f};
    return v * sv_f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

constexpr int SV_PH = 2 * SV_EH + 1, SV_PW = 2 * SV_EW + 1, SV_PP = SV_PW + 1;     // image patch rows / columns / row pitch
static_assert(SV_EH * SV_EW == 256, "one thread per stem pixel of the tile's region");

__global__ __launch_bounds__(256) void stem_dw_valu_kernel(const float* __restrict__ x, const float* __restrict__ wk,
                                                            const float* __restrict__ ss, const float* __restrict__ sb,
                                                            const float* __restrict__ wd, const float* __restrict__ dsc,
                                                            const float* __restrict__ dbi, float* __restrict__ y,
                                                            float* __restrict__ partial, int img_h, int img_w, int Hs, int Ws,
                                                            int stem_pad_t, int stem_pad_l, int pad_t, int pad_l, int Ho, int Wo,
                                                            int tiles_w) {
    __shared__ __attribute__((aligned(16))) float E[SV_EH * SV_EW * SV_LDE];
    __shared__ float X[3 * SV_PH * SV_PP];       // the image patch under the tile's stem region: [plane][33 rows][33 (+1) columns]
    __shared__ sv_f32x4 red[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, tile = blockIdx.x;
    const int th_i = tile / tiles_w;
    const int ho0 = th_i * SV_TH, wo0 = (tile - th_i * tiles_w) * SV_TW;

    // ---- the image patch: every element requested at once (one memory latency per tile), zeros outside the image
    {
        const size_t plane = (size_t)img_h * img_w;
        const float* xb = x + (size_t)b * 3 * plane;
        const int ih0 = (ho0 - pad_t) * 2 - stem_pad_t, iw0 = (wo0 - pad_l) * 2 - stem_pad_l;
        // columns 0 .. 31: a wave-load takes two patch rows (lanes 0-31 | 32-63), the four waves share a plane's 17 row pairs;
        // 32-bit element offsets from the image's first plane (an image has far fewer than 2^31 elements)
        const int pc = lane & 31, half = lane >> 5;
        const int iw = iw0 + pc;
        const bool col_ok = (unsigned)iw < (unsigned)img_w;
        const int plane_i = img_h * img_w;
        float v[3][5];
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int pr = 2 * (wave + 4 * jj) + half;
            const int ih = ih0 + pr;
            const bool ok = col_ok && pr < SV_PH && (unsigned)ih < (unsigned)img_h;
            const int off = ok ? ih * img_w + iw : 0;                      // a select on the address, never a branch
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const float t = xb[off + (ok ? pl * plane_i : 0)];
                v[pl][jj] = ok ? t : 0.f;
            }
        }
        // column 32: one element per thread
        const int lpl = tid / SV_PH, lpr = tid - lpl * SV_PH;
        float vl;
        {
            const int ih = ih0 + lpr, iwl = iw0 + SV_PW - 1;
            const bool ok = tid < 3 * SV_PH && (unsigned)ih < (unsigned)img_h && (unsigned)iwl < (unsigned)img_w;
            const float t = xb[ok ? lpl * plane_i + ih * img_w + iwl : 0];
            vl = ok ? t : 0.f;
        }
        float* xrow = X + (2 * wave + half) * SV_PP + pc;
#pragma unroll
        for (int jj = 0; jj < 5; ++jj) {
            const int pr = 2 * (wave + 4 * jj) + half;
            if (pr < SV_PH) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) xrow[(pl * SV_PH + 8 * jj) * SV_PP] = v[pl][jj];
            }
        }
        if (tid < 3 * SV_PH) X[(lpl * SV_PH + lpr) * SV_PP + SV_PW - 1] = vl;
    }
    __syncthreads();

    // ---- stem: this thread's pixel of the 16 x 16 region
    {
        const int sy = tid >> 4, sx = tid & 15;
        const int hs = ho0 - pad_t + sy, wsx = wo0 - pad_l + sx;
        const bool inside = (unsigned)hs < (unsigned)Hs && (unsigned)wsx < (unsigned)Ws;
        sv_f32x2 acc[SV_C / 2];
#pragma unroll
        for (int c2 = 0; c2 < SV_C / 2; ++c2) acc[c2] = sv_f32x2{0.f, 0.f};
        const sv_f32x2* wk2 = reinterpret_cast<const sv_f32x2*>(wk);              // [27][16] channel pairs, uniform addresses
        const float* xp = X + (2 * sy) * SV_PP + 2 * sx;
        // one filter tap per trip (not unrolled: the 96 weights of a tap fit the scalar registers, all 864 would not)
        float xn[3];
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) xn[ci] = xp[ci * SV_PH * SV_PP];
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // the next tap's three values are read while this tap multiplies (the last trip re-reads tap 8)
            const float xv[3] = {xn[0], xn[1], xn[2]};
            const int tn = tap < 8 ? tap + 1 : 8;
            const int kh = tn / 3, kw = tn - kh * 3;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) xn[ci] = xp[(ci * SV_PH + kh) * SV_PP + kw];
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const sv_f32x2 xk = {xv[ci], xv[ci]};
                const sv_f32x2* wrow = wk2 + (tap * 3 + ci) * (SV_C / 2);
#pragma unroll
                for (int c2 = 0; c2 < SV_C / 2; ++c2) acc[c2] = __builtin_elementwise_fma(xk, wrow[c2], acc[c2]);
            }
        }
        const sv_f32x2* ss2 = reinterpret_cast<const sv_f32x2*>(ss);
        const sv_f32x2* sb2 = reinterpret_cast<const sv_f32x2*>(sb);
        float* erow = E + tid * SV_LDE;
        const float keep = inside ? 1.f : 0.f;
        // BN + swish + store; one uniform branch on "both BN vectors given" (the usual case) instead of two per channel pair
        auto finish = [&](auto both) __attribute__((always_inline)) {
#pragma unroll
            for (int c4 = 0; c4 < SV_C / 4; ++c4) {
                sv_f32x4 o;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c2 = 2 * c4 + h;
                    sv_f32x2 v = acc[c2];
                    if constexpr (decltype(both)::value) {
                        v = __builtin_elementwise_fma(v, ss2[c2], sb2[c2]);
                    } else {
                        if (ss) v = v * ss2[c2];
                        if (sb) v = v + sb2[c2];
                    }
                    v = sv_swish2(v * sv_f32x2{keep, keep});          // masked by a multiply, not a branch around the transcendentals
                    o[2 * h] = v[0];
                    o[2 * h + 1] = v[1];
                }
                *reinterpret_cast<sv_f32x4*>(erow + 4 * c4) = o;
            }
        };
        if (ss && sb) finish(std::true_type{});
        else finish(std::false_type{});
    }
    __syncthreads();

    // ---- depthwise 3x3: strip st (row r, 7 columns) x channel quad q
    const int q = tid & 7, st = tid >> 3;
    sv_f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    if (st < SV_TH * (SV_TW / SV_SW)) {
        const int r = st >> 1, col0 = (st & 1) * SV_SW;
        sv_f32x4 w9[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w9[t] = *reinterpret_cast<const sv_f32x4*>(wd + t * SV_C + 4 * q);
        sv_f32x4 o[SV_SW];
#pragma unroll
        for (int j = 0; j < SV_SW; ++j) o[j] = sv_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < SV_SW + 2; ++dx) {
                const sv_f32x4 e = *reinterpret_cast<const sv_f32x4*>(E + ((r + dy) * SV_EW + col0 + dx) * SV_LDE + 4 * q);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int j = dx - kx;
                    if (j >= 0 && j < SV_SW) o[j] = __builtin_elementwise_fma(e, w9[dy * 3 + kx], o[j]);
                }
            }
        sv_f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (dsc) sc = *reinterpret_cast<const sv_f32x4*>(dsc + 4 * q);
        if (dbi) bi = *reinterpret_cast<const sv_f32x4*>(dbi + 4 * q);
        const int ho = ho0 + r;
        if (ho0 + SV_TH <= Ho && wo0 + SV_TW <= Wo) {          // whole tile inside the output (uniform): no masks
            float* yrow = y + (((size_t)b * Ho + ho) * Wo + wo0 + col0) * SV_C + 4 * q;
#pragma unroll
            for (int j = 0; j < SV_SW; ++j) {
                const sv_f32x4 t = o[j] * sc + bi;
                const sv_f32x2 lo = sv_swish2(sv_f32x2{t[0], t[1]}), hi = sv_swish2(sv_f32x2{t[2], t[3]});
                const sv_f32x4 v = {lo[0], lo[1], hi[0], hi[1]};
                __builtin_nontemporal_store(v, reinterpret_cast<sv_f32x4*>(yrow + (size_t)j * SV_C));   // streaming: read next by another kernel
                sum += v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SV_SW; ++j) {
                const int wo = wo0 + col0 + j;
                sv_f32x4 v = o[j] * sc + bi;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = v[t] * sigmoid_fast(v[t]);
                const bool valid = ho < Ho && wo < Wo;
                if (valid) __builtin_nontemporal_store(v, reinterpret_cast<sv_f32x4*>(y + (((size_t)b * Ho + ho) * Wo + wo) * SV_C + 4 * q));
                const float m = valid ? 1.f : 0.f;
                sum += v * m;
            }
        }
    }
    if (partial) {
        // per-channel sum over the tile in a fixed order: lanes with equal q inside a wave, then the four waves
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            sum[0] += __shfl_xor(sum[0], off, 64); sum[1] += __shfl_xor(sum[1], off, 64);
            sum[2] += __shfl_xor(sum[2], off, 64); sum[3] += __shfl_xor(sum[3], off, 64);
        }
        if (lane < 8) red[wave][lane] = sum;
        __syncthreads();
        if (tid < 8) {
            const sv_f32x4 t = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            *reinterpret_cast<sv_f32x4*>(partial + ((size_t)b * gridDim.x + tile) * SV_C + 4 * tid) = t;
        }
    }
}

// 1 when the vector-ALU stem kernel runs this layer: 32 stem channels (its chunks per image = icap_stem_dwconv_chunks)
int stem_dwconv_valu_supported(int C) { return C == SV_C; }

// x: planar fp32 image [B][3][img_h][img_w]; wk: the stem weights k-major, [27][C] with k = (kh * 3 + kw) * 3 + input channel
int stem_dwconv_valu_f32(const float* x, const float* wk, const float* ss, const float* sb, const float* wd, const float* ds,
                         const float* db, float* y, float* partial, int B, int img_h, int img_w, int Hs, int Ws, int C,
                         int stem_pad_t, int stem_pad_l, int pad_t, int pad_l, int Ho, int Wo, hipStream_t st) {
    if (!x || !wk || !wd || !y || B <= 0 || img_h <= 0 || img_w <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || stem_pad_t < 0 ||
        stem_pad_l < 0 || pad_t < 0 || pad_l < 0) {
        set_error("stem_dwconv_valu: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (C != SV_C) { set_error("stem_dwconv_valu: C=%d, the kernel is built for %d stem channels", C, SV_C); return ICAP_ERR_ARG; }
    if (pad_t > 1 || pad_l > 1) { set_error("stem_dwconv_valu: depthwise padding %d / %d exceeds the tile's halo of 1", pad_t, pad_l); return ICAP_ERR_ARG; }
    if ((long long)(Hs - 1) * 2 - stem_pad_t >= img_h || (long long)(Ws - 1) * 2 - stem_pad_l >= img_w ||
        (long long)(Ho - 1) - pad_t >= Hs || (long long)(Wo - 1) - pad_l >= Ws) {
        set_error("stem_dwconv_valu: output sizes do not fit the image");
        return ICAP_ERR_ARG;
    }
    if (!aligned16(x) || !aligned16(wk) || !aligned16(wd) || !aligned16(y) || (partial && !aligned16(partial)) || (ss && !aligned16(ss)) ||
        (sb && !aligned16(sb)) || (ds && !aligned16(ds)) || (db && !aligned16(db))) {
        set_error("stem_dwconv_valu: pointers must be 16-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    if (B > 65535) { set_error("stem_dwconv_valu: B=%d exceeds 65535", B); return ICAP_ERR_LIMIT; }
    const int tiles_w = (Wo + SV_TW - 1) / SV_TW;
    const int ntiles = ((Ho + SV_TH - 1) / SV_TH) * tiles_w;
    hipLaunchKernelGGL(stem_dw_valu_kernel, dim3(ntiles, B), dim3(256), 0, st, x, wk, ss, sb, wd, ds, db, y, partial, img_h, img_w, Hs, Ws,
                       stem_pad_t, stem_pad_l, pad_t, pad_l, Ho, Wo, tiles_w);
    return launch_status("stem_dwconv_valu");
}

}  // namespace icap
