// Bandwidth-bound spatial kernels of the CNN encoders (NHWC fp32):
// layout change for the network input, depthwise convolution with fused
// batch-norm + activation and deterministic partial pooling for squeeze-excite,
// the squeeze-excite gate itself, max pooling and average pooling.
// All of them move 16 bytes per lane per access along the channel axis.
#include <stdlib.h>
#include "common.hpp"

namespace icap {

// ---------------------------------------------------------------- layout
// NCHW fp32 image -> NHWC with the channel axis zero-padded to Cp (Cp % 4 == 0),
// so the first convolution can stage 16-byte channel groups like every other one.
__global__ void nchw_to_nhwc_pad_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                        int C, int HW, int Cp, long long total_pix) {
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*HW
    if (pix >= total_pix) return;
    const long long b = pix / HW;
    const int s = (int)(pix - b * HW);
    const float* sp = src + (size_t)b * C * HW + s;
    float* dp = dst + (size_t)pix * Cp;
    for (int c = 0; c < Cp; c += 4) {
        float4 v;
        v.x = (c + 0 < C) ? sp[(size_t)(c + 0) * HW] : 0.f;
        v.y = (c + 1 < C) ? sp[(size_t)(c + 1) * HW] : 0.f;
        v.z = (c + 2 < C) ? sp[(size_t)(c + 2) * HW] : 0.f;
        v.w = (c + 3 < C) ? sp[(size_t)(c + 3) * HW] : 0.f;
        *reinterpret_cast<float4*>(dp + c) = v;
    }
}

int nchw_to_nhwc_pad_f32(const float* src, float* dst, int B, int C, int H, int W, int Cp, hipStream_t st) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C || Cp % 4) {
        set_error("nchw_to_nhwc_pad: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (!aligned16(dst)) { set_error("nchw_to_nhwc_pad: dst must be 16-byte aligned"); return ICAP_ERR_ALIGN; }
    const long long total = (long long)B * H * W;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) { set_error("nchw_to_nhwc_pad: too large"); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, C, H * W, Cp, total);
    return launch_status("nchw_to_nhwc_pad");
}

// ---------------------------------------------------------------- depthwise conv
// Register-strip kernel: a thread owns one channel quad (16 B of the NHWC channel axis) and a strip of TW
// consecutive output pixels of one output row.  For every filter row it walks the input row once, each
// loaded float4 feeding all outputs whose window covers it, so a strip costs ((TW-1)*S+KS)*KS loads instead
// of TW*KS*KS; rows shared by vertically adjacent strips come from L1/L2.  Threads are laid out with the
// channel quad fastest (coalesced 16-B lanes) and a workgroup covers `spb` whole strips of one image.
// The per-channel sums of the workgroup's outputs (needed by squeeze-excite) are reduced through LDS in a
// fixed order and written to partial[b][workgroup][C]; the gate kernel adds the workgroups in order, so the
// pooled value does not depend on scheduling.
constexpr int DW_TW = 7;   // divides every EfficientNet-B0 feature-map width (112, 56, 28, 14, 7)

template <int KS, int S, int TH>
__global__ __launch_bounds__(512) void dwconv_strip_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
    const float* __restrict__ bias, float* __restrict__ y, float* __restrict__ partial, int H, int W,
    int C, int pad_t, int pad_l, int Ho, int Wo, int act, int nq, int spb, int nstrips_w,
    float* __restrict__ stat_part = nullptr, float* __restrict__ stat_cnt = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float4 red[];
    constexpr int TW = DW_TW;
    constexpr int NIN = (TW - 1) * S + KS;     // input columns feeding one strip
    constexpr int NROW = (TH - 1) * S + KS;    // input rows feeding TH vertically adjacent strips
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int ls = tid / nq;
    const int cq = tid - ls * nq;
    const int strip = blockIdx.x * spb + ls;                 // a "strip" is TH output rows x TW columns
    const int nstrips = ((Ho + TH - 1) / TH) * nstrips_w;
    const bool active = (ls < spb) && (strip < nstrips);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sq = make_float4(0.f, 0.f, 0.f, 0.f);    // stat_part: squared deviations of this thread's outputs about their mean
    float nvalid = 0.f;
    if (active) {
        const int c = cq << 2;
        const int hs = strip / nstrips_w;
        const int ho0 = hs * TH;
        const int wo0 = (strip - hs * nstrips_w) * TW;
        const int hi0 = ho0 * S - pad_t, wi0 = wo0 * S - pad_l;
        const float* xb = x + (size_t)b * H * W * C + c;
        float4 acc[TH][TW];
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int o = 0; o < TW; ++o) acc[t][o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ri = 0; ri < NROW; ++ri) {
            const int hi = hi0 + ri;
            const bool row_ok = (unsigned)hi < (unsigned)H;   // wave-divergent only at image borders
            if (!row_ok) continue;
            const float* xr = xb + (size_t)hi * W * C;
            float4 v[NIN];
#pragma unroll
            for (int j = 0; j < NIN; ++j) {
                const int wi = wi0 + j;
                const bool ok = (unsigned)wi < (unsigned)W;
                v[j] = *reinterpret_cast<const float4*>(ok ? xr + (size_t)wi * C : xr);   // always a valid address
                if (!ok) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int t = 0; t < TH; ++t) {
                const int r = ri - t * S;                     // filter row this input row is for output row ho0 + t
                if (r < 0 || r >= KS) continue;
                float4 wr[KS];
#pragma unroll
                for (int q = 0; q < KS; ++q) wr[q] = *reinterpret_cast<const float4*>(w + (size_t)(r * KS + q) * C + c);
#pragma unroll
                for (int j = 0; j < NIN; ++j)
#pragma unroll
                    for (int o = 0; o < TW; ++o) {
                        const int q = j - o * S;
                        if (q >= 0 && q < KS) {
                            acc[t][o].x = fmaf(v[j].x, wr[q].x, acc[t][o].x); acc[t][o].y = fmaf(v[j].y, wr[q].y, acc[t][o].y);
                            acc[t][o].z = fmaf(v[j].z, wr[q].z, acc[t][o].z); acc[t][o].w = fmaf(v[j].w, wr[q].w, acc[t][o].w);
                        }
                    }
            }
        }
        const float4 sc = scale ? *reinterpret_cast<const float4*>(scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 bi = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        dispatch_act(act, [&](auto tag) {
            constexpr int ACT = decltype(tag)::value;
#pragma unroll
            for (int t = 0; t < TH; ++t) {
                const int ho = ho0 + t;
                if (ho >= Ho) continue;
                float* yr = y + ((size_t)b * Ho * Wo + (size_t)ho * Wo) * C + c;
#pragma unroll
                for (int o = 0; o < TW; ++o) {
                    const int wo = wo0 + o;
                    if (wo < Wo) {
                        float4 u;
                        u.x = act_c<ACT>(fmaf(acc[t][o].x, sc.x, bi.x)); u.y = act_c<ACT>(fmaf(acc[t][o].y, sc.y, bi.y));
                        u.z = act_c<ACT>(fmaf(acc[t][o].z, sc.z, bi.z)); u.w = act_c<ACT>(fmaf(acc[t][o].w, sc.w, bi.w));
                        *reinterpret_cast<float4*>(yr + (size_t)wo * C) = u;
                        sum.x += u.x; sum.y += u.y; sum.z += u.z; sum.w += u.w;
                        acc[t][o] = u;
                        nvalid += 1.f;
                    }
                }
            }
        });
        if (stat_part && nvalid > 0.f) {
            const float inv = 1.f / nvalid;
            const float4 mu = make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv);
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int o = 0; o < TW; ++o)
                    if (ho0 + t < Ho && wo0 + o < Wo) {
                        const float dx = acc[t][o].x - mu.x, dy = acc[t][o].y - mu.y, dz = acc[t][o].z - mu.z, dw = acc[t][o].w - mu.w;
                        sq.x = fmaf(dx, dx, sq.x); sq.y = fmaf(dy, dy, sq.y); sq.z = fmaf(dz, dz, sq.z); sq.w = fmaf(dw, dw, sq.w);
                    }
        }
    }
    if (stat_part) {
        // per-workgroup statistics of the outputs for a train-mode BatchNorm: the strips of a channel quad are merged in
        // strip order; red holds [sum | squared deviations | count] per thread
        float4* redq = red + blockDim.x;
        float* redn = reinterpret_cast<float*>(red + 2 * blockDim.x);
        red[tid] = sum;
        redq[tid] = sq;
        redn[tid] = nvalid;
        __syncthreads();
        if (tid < nq) {
            ColStat a[4] = {{nvalid, sum.x, sq.x}, {nvalid, sum.y, sq.y}, {nvalid, sum.z, sq.z}, {nvalid, sum.w, sq.w}};
            for (int l = 1; l < spb; ++l) {
                const float4 so = red[l * nq + tid], qo = redq[l * nq + tid];
                const float no = redn[l * nq + tid];
                stat_merge(a[0], no, so.x, qo.x); stat_merge(a[1], no, so.y, qo.y);
                stat_merge(a[2], no, so.z, qo.z); stat_merge(a[3], no, so.w, qo.w);
            }
            const size_t pidx = (size_t)b * gridDim.x + blockIdx.x;
            *reinterpret_cast<float4*>(stat_part + (pidx * 2 + 0) * C + (tid << 2)) = make_float4(a[0].s, a[1].s, a[2].s, a[3].s);
            *reinterpret_cast<float4*>(stat_part + (pidx * 2 + 1) * C + (tid << 2)) = make_float4(a[0].q, a[1].q, a[2].q, a[3].q);
            if (tid == 0) stat_cnt[pidx] = a[0].n;
        }
        return;
    }
    if (partial) {
        red[tid] = sum;
        __syncthreads();
        if (tid < nq) {
            float4 t = red[tid];
            for (int l = 1; l < spb; ++l) {
                const float4 o = red[l * nq + tid];
                t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
            }
            *reinterpret_cast<float4*>(partial + ((size_t)b * gridDim.x + blockIdx.x) * C + (tid << 2)) = t;
        }
    }
}

// Rows per thread: two vertically adjacent strips share (KS - S) input rows in registers, which matters when
// the filter is large against the stride; maps with an odd number of rows keep one row per thread.
static int dwconv_th(int Ho, int Wo, int C, int KS, int stride) {
    const long long threads2 = (long long)((Ho + 1) / 2) * ((Wo + DW_TW - 1) / DW_TW) * (C / 4);
    (void)KS; (void)stride;
    // measured at batch 1024: even row counts from 14 up gain 20-27 % from the shared rows, 7-row maps lose 10-35 %
    return (threads2 >= 2048 || (Ho >= 14 && Ho % 2 == 0)) ? 2 : 1;
}

// Launch geometry shared by the kernel launcher and by callers sizing the partial buffer.
static void dwconv_geometry(int Ho, int Wo, int C, int th, int* block, int* spb, int* nblk) {
    const int nq = C / 4;
    int best_bd = 0, best_spb = 1;
    double best_u = -1.0;
    for (int bd = 128; bd <= 512; bd += 64) {
        if (bd < nq) continue;
        const int s = bd / nq;
        const double u = (double)(s * nq) / bd;
        if (u > best_u + 1e-9) { best_u = u; best_bd = bd; best_spb = s; }
    }
    const int nstrips = ((Ho + th - 1) / th) * ((Wo + DW_TW - 1) / DW_TW);
    if (best_spb > nstrips) best_spb = nstrips;
    *block = best_bd; *spb = best_spb; *nblk = (nstrips + best_spb - 1) / best_spb;
}

int dwconv_chunks(int Ho, int Wo, int C, int KS, int stride) {
    if (Ho <= 0 || Wo <= 0 || C <= 0 || C % 4 || C / 4 > 512) return -1;
    int bd, spb, nblk;
    dwconv_geometry(Ho, Wo, C, dwconv_th(Ho, Wo, C, KS, stride), &bd, &spb, &nblk);
    return nblk;
}

static int dwconv2d_launch(const float* x, const float* w, const float* scale, const float* bias, float* y,
                           float* partial, int B, int H, int W, int C, int KS, int stride, int pad_t, int pad_l,
                           int Ho, int Wo, int act, float* stat_part, float* stat_cnt, hipStream_t st) {
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || stride <= 0 || Ho <= 0 || Wo <= 0) {
        set_error("dwconv2d: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (C % 4) { set_error("dwconv2d: C=%d must be a multiple of 4", C); return ICAP_ERR_ALIGN; }
    if (C / 4 > 512) { set_error("dwconv2d: C=%d exceeds 2048", C); return ICAP_ERR_LIMIT; }
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (partial && !aligned16(partial)) ||
        (scale && !aligned16(scale)) || (bias && !aligned16(bias))) {
        set_error("dwconv2d: pointers must be 16-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    if (B > 65535) { set_error("dwconv2d: B=%d exceeds 65535", B); return ICAP_ERR_LIMIT; }
    int bd, spb, nblk;
    const int th = dwconv_th(Ho, Wo, C, KS, stride);
    dwconv_geometry(Ho, Wo, C, th, &bd, &spb, &nblk);
    const int nq = C / 4, nsw = (Wo + DW_TW - 1) / DW_TW;
    const dim3 grid(nblk, B), block(bd);
    const size_t lds = stat_part ? (size_t)bd * (2 * sizeof(float4) + sizeof(float)) : (partial ? (size_t)bd * sizeof(float4) : 0);
#define ICAP_DW(K_, S_)                                                                                                   \
    do {                                                                                                                 \
        if (th == 2) hipLaunchKernelGGL((dwconv_strip_kernel<K_, S_, 2>), grid, block, lds, st, x, w, scale, bias, y, partial, H, W, C, pad_t, pad_l, Ho, Wo, act, nq, spb, nsw, stat_part, stat_cnt); \
        else hipLaunchKernelGGL((dwconv_strip_kernel<K_, S_, 1>), grid, block, lds, st, x, w, scale, bias, y, partial, H, W, C, pad_t, pad_l, Ho, Wo, act, nq, spb, nsw, stat_part, stat_cnt); \
    } while (0)
    if (KS == 3 && stride == 1) ICAP_DW(3, 1);
    else if (KS == 3 && stride == 2) ICAP_DW(3, 2);
    else if (KS == 5 && stride == 1) ICAP_DW(5, 1);
    else if (KS == 5 && stride == 2) ICAP_DW(5, 2);
    else { set_error("dwconv2d: kernel %d stride %d not supported (3 or 5, stride 1 or 2)", KS, stride); return ICAP_ERR_ARG; }
#undef ICAP_DW
    return launch_status("dwconv2d");
}

int dwconv2d_f32(const float* x, const float* w, const float* scale, const float* bias, float* y,
                 float* partial, int B, int H, int W, int C, int KS, int stride, int pad_t, int pad_l,
                 int Ho, int Wo, int act, hipStream_t st) {
    return dwconv2d_launch(x, w, scale, bias, y, partial, B, H, W, C, KS, stride, pad_t, pad_l, Ho, Wo, act, nullptr, nullptr, st);
}

// Plain depthwise convolution that also leaves the per-workgroup statistics of its result for a train-mode BatchNorm:
// stat_part [P][2][C] (sum, squared deviations about the workgroup's own mean) and stat_cnt [P] (outputs per channel in the
// workgroup), P = B * dwconv_chunks(Ho, Wo, C, KS, stride).
int dwconv2d_stats_f32(const float* x, const float* w, float* y, int B, int H, int W, int C, int KS, int stride, int pad_t,
                       int pad_l, int Ho, int Wo, float* stat_part, float* stat_cnt, hipStream_t st) {
    if (!stat_part || !stat_cnt || !aligned16(stat_part)) { set_error("dwconv2d_stats: null or unaligned statistics buffers"); return ICAP_ERR_ARG; }
    return dwconv2d_launch(x, w, nullptr, nullptr, y, nullptr, B, H, W, C, KS, stride, pad_t, pad_l, Ho, Wo, ACT_NONE, stat_part, stat_cnt, st);
}

// ---------------------------------------------------------------- squeeze-excite gate
// A workgroup owns SE_IMG images: pooled = sum of chunk partials / HW; s = swish(Wr pooled + br);
// gate = sigmoid(We s + be).  Both weight matrices are stored [S][C] (the expand weight transposed) so that
// every access is contiguous along the channel axis.  Every weight element loaded is used for all SE_IMG images:
// one image per workgroup re-read both matrices from L2 for each image (0.44 MB per image at C = 1152, S = 48,
// 1.1 GB per launch at B = 1024 with the slices below).
constexpr int SE_MAX_C = 2048;
constexpr int SE_MAX_S = 128;
constexpr int SE_IMG = 4;
constexpr int SE_THREADS = 256 * SE_IMG;      // 256 threads pool one image each; all of them share the two matrix products

__global__ __launch_bounds__(SE_THREADS) void se_gate_kernel(
    const float* __restrict__ partial, int nchunks, float inv_hw, const float* __restrict__ wr,
    const float* __restrict__ br, const float* __restrict__ we, const float* __restrict__ be,
    float* __restrict__ gate, int B, int C, int S) {
    __shared__ __attribute__((aligned(16))) float pooled[SE_IMG][SE_MAX_C];
    __shared__ float sq[SE_IMG][SE_MAX_S];
    __shared__ __attribute__((aligned(16))) float4 red[SE_IMG][256];
    const int b0 = blockIdx.x * SE_IMG, tid = threadIdx.x;
    const int nimg = min(SE_IMG, B - b0);
    const int nq = C >> 2;
    // Pool: sum the chunk partials of every channel quad.  The 256 threads cover (quad, chunk lane) pairs;
    // with few channels and many chunks (early layers) several lanes split the chunk axis (interleaved) and
    // are combined through LDS in a fixed order, so the sum is independent of scheduling.
    int kl = 1;
    while (kl * 2 * nq <= 256 && kl * 2 <= nchunks) kl *= 2;
    {
        const int im = tid >> 8, t = tid & 255;            // image of this 256-thread group, thread within the group
        const float* pb = partial + (size_t)(b0 + min(im, nimg - 1)) * nchunks * C;
        for (int q0 = 0; q0 < nq; q0 += 256 / kl) {
            const int ql = t % (256 / kl), l = t / (256 / kl);
            const int q = q0 + ql;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < nq) {
#pragma unroll 4
                for (int k = l; k < nchunks; k += kl) {
                    const float4 v = *reinterpret_cast<const float4*>(pb + (size_t)k * C + (q << 2));
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            }
            if (kl > 1) {
                red[im][t] = s;
                __syncthreads();
                if (l == 0 && q < nq) {
                    for (int j = 1; j < kl; ++j) {
                        const float4 o = red[im][j * (256 / kl) + ql];
                        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
                    }
                }
                __syncthreads();
            }
            if (l == 0 && q < nq) {
                s.x *= inv_hw; s.y *= inv_hw; s.z *= inv_hw; s.w *= inv_hw;
                *reinterpret_cast<float4*>(&pooled[im][q << 2]) = s;
            }
        }
    }
    __syncthreads();
    // Squeeze: one wave per output j, 16-byte weight loads, each used for all images of the workgroup.
    const int wave = tid >> 6, lane = tid & 63;
    for (int j = wave; j < S; j += SE_THREADS / 64) {
        float a[SE_IMG];
#pragma unroll
        for (int im = 0; im < SE_IMG; ++im) a[im] = 0.f;
        const float* wj = wr + (size_t)j * C;
#pragma unroll 2
        for (int q = lane; q < nq; q += 64) {
            const float4 wv = *reinterpret_cast<const float4*>(wj + (q << 2));
#pragma unroll
            for (int im = 0; im < SE_IMG; ++im) a[im] = dot4(wv, *reinterpret_cast<const float4*>(&pooled[im][q << 2]), a[im]);
        }
        const float bj = br[j];
#pragma unroll
        for (int im = 0; im < SE_IMG; ++im) {
            const float t = wave_sum(a[im]);
            if (lane == 0) sq[im][j] = act_apply(t + bj, ACT_SWISH);
        }
    }
    __syncthreads();
    // Excite: this workgroup's slice of the channel quads (gridDim.y slices per image group); a thread owns one
    // (image, quad) pair, so that the 1024 threads have work even when the slice is a few hundred quads wide.
    const int per = (nq + gridDim.y - 1) / gridDim.y;
    const int q_lo = blockIdx.y * per, q_hi = min(nq, q_lo + per), nqs = max(q_hi - q_lo, 0);
    for (int wi = tid; wi < nimg * nqs; wi += SE_THREADS) {
        const int im = wi / nqs, q = q_lo + (wi - im * nqs);
        float4 a = *reinterpret_cast<const float4*>(be + (q << 2));
#pragma unroll 8
        for (int j = 0; j < S; ++j) {
            const float4 wv = *reinterpret_cast<const float4*>(we + (size_t)j * C + (q << 2));
            const float sj = sq[im][j];
            a.x = fmaf(wv.x, sj, a.x); a.y = fmaf(wv.y, sj, a.y); a.z = fmaf(wv.z, sj, a.z); a.w = fmaf(wv.w, sj, a.w);
        }
        a.x = sigmoidf_(a.x); a.y = sigmoidf_(a.y); a.z = sigmoidf_(a.z); a.w = sigmoidf_(a.w);
        *reinterpret_cast<float4*>(gate + (size_t)(b0 + im) * C + (q << 2)) = a;
    }
}

int se_gate_f32(const float* partial, int nchunks, float inv_hw, const float* wr, const float* br,
                const float* we, const float* be, float* gate, int B, int C, int S, hipStream_t st) {
    if (!partial || !wr || !br || !we || !be || !gate || B <= 0 || C <= 0 || S <= 0 || nchunks <= 0) {
        set_error("se_gate: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (C > SE_MAX_C || S > SE_MAX_S) { set_error("se_gate: C=%d S=%d exceed %d/%d", C, S, SE_MAX_C, SE_MAX_S); return ICAP_ERR_LIMIT; }
    if (C % 4 || !aligned16(partial) || !aligned16(wr) || !aligned16(we) || !aligned16(be) || !aligned16(gate)) {
        set_error("se_gate: C %% 4 and 16-byte alignment required");
        return ICAP_ERR_ALIGN;
    }
    const int groups = (B + SE_IMG - 1) / SE_IMG;
    // several workgroups per image group only when the groups alone leave CUs idle (each recomputes the pooled / squeeze
    // vectors and writes its own slice of the gate)
    int slices = 1;
    while (slices < 8 && groups * slices < 256 && (C >> 2) / (slices * 2) >= 32) slices *= 2;
    hipLaunchKernelGGL(se_gate_kernel, dim3(groups, slices), dim3(SE_THREADS), 0, st, partial, nchunks, inv_hw, wr, br, we, be, gate, B, C, S);
    return launch_status("se_gate");
}

// ---------------------------------------------------------------- max pooling
__global__ void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C,
                               int KS, int stride, int pad, int Ho, int Wo, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*Ho*Wo*(C/4)
    if (i >= total) return;
    const int nq = C >> 2;
    const int cq = (int)(i % nq);
    long long pix = i / nq;
    const int wo = (int)(pix % Wo); pix /= Wo;
    const int ho = (int)(pix % Ho);
    const long long b = pix / Ho;
    const float* xb = x + (size_t)b * H * W * C + (cq << 2);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int r = 0; r < KS; ++r) {
        const int hi = ho * stride - pad + r;
        if ((unsigned)hi >= (unsigned)H) continue;
        for (int q = 0; q < KS; ++q) {
            const int wi = wo * stride - pad + q;
            if ((unsigned)wi >= (unsigned)W) continue;
            const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)hi * W + wi) * C);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    *reinterpret_cast<float4*>(y + (((size_t)b * Ho + ho) * Wo + wo) * C + (cq << 2)) = m;
}

int maxpool2d_f32(const float* x, float* y, int B, int H, int W, int C, int KS, int stride, int pad,
                  int Ho, int Wo, hipStream_t st) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || KS <= 0 || stride <= 0 || pad < 0 || Ho <= 0 || Wo <= 0) {
        set_error("maxpool2d: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (C % 4 || !aligned16(x) || !aligned16(y)) { set_error("maxpool2d: C %% 4 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    if (pad * 2 > KS) { set_error("maxpool2d: pad larger than half the window"); return ICAP_ERR_ARG; }
    const long long total = (long long)B * Ho * Wo * (C / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) { set_error("maxpool2d: too large"); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, H, W, C, KS, stride, pad, Ho, Wo, total);
    return launch_status("maxpool2d");
}

// bf16 activations (kept between the dense layers in bf16 compute mode): eight channels per lane.  The maximum of bf16
// values is one of them, so the result is exact.
__global__ void maxpool_bf16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int H, int W, int C,
                                    int KS, int stride, int pad, int Ho, int Wo, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*Ho*Wo*(C/8)
    if (i >= total) return;
    const int nq = C >> 3;
    const int cq = (int)(i % nq);
    long long pix = i / nq;
    const int wo = (int)(pix % Wo); pix /= Wo;
    const int ho = (int)(pix % Ho);
    const long long b = pix / Ho;
    const unsigned short* xb = x + (size_t)b * H * W * C + (cq << 3);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int r = 0; r < KS; ++r) {
        const int hi = ho * stride - pad + r;
        if ((unsigned)hi >= (unsigned)H) continue;
        for (int q = 0; q < KS; ++q) {
            const int wi = wo * stride - pad + q;
            if ((unsigned)wi >= (unsigned)W) continue;
            const uint4 v = *reinterpret_cast<const uint4*>(xb + ((size_t)hi * W + wi) * C);
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m[2 * j] = fmaxf(m[2 * j], __uint_as_float(u[j] << 16));
                m[2 * j + 1] = fmaxf(m[2 * j + 1], __uint_as_float(u[j] & 0xffff0000u));
            }
        }
    }
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (__float_as_uint(m[2 * j]) >> 16) | (__float_as_uint(m[2 * j + 1]) & 0xffff0000u);
    *reinterpret_cast<uint4*>(y + (((size_t)b * Ho + ho) * Wo + wo) * C + (cq << 3)) = make_uint4(o[0], o[1], o[2], o[3]);
}

int maxpool2d_bf16(const unsigned short* x, unsigned short* y, int B, int H, int W, int C, int KS, int stride, int pad, int Ho, int Wo,
                   hipStream_t st) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || KS <= 0 || stride <= 0 || pad < 0 || Ho <= 0 || Wo <= 0) {
        set_error("maxpool2d_bf16: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (C % 8 || !aligned16(x) || !aligned16(y)) { set_error("maxpool2d_bf16: C %% 8 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    if (pad * 2 > KS) { set_error("maxpool2d_bf16: pad larger than half the window"); return ICAP_ERR_ARG; }
    const long long total = (long long)B * Ho * Wo * (C / 8);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) { set_error("maxpool2d_bf16: too large"); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(maxpool_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, H, W, C, KS, stride, pad, Ho, Wo, total);
    return launch_status("maxpool2d_bf16");
}

// ---------------------------------------------------------------- average pooling
// Adaptive average pooling NHWC -> [B, OH, OW, C] with torch's window rule
// start = floor(i*H/OH), end = ceil((i+1)*H/OH).  OH = OW = 1 is global pooling.
__global__ void adaptive_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                        int C, int OH, int OW, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*OH*OW*(C/4)
    if (i >= total) return;
    const int nq = C >> 2;
    const int cq = (int)(i % nq);
    long long pix = i / nq;
    const int ow = (int)(pix % OW); pix /= OW;
    const int oh = (int)(pix % OH);
    const long long b = pix / OH;
    const int h0 = (oh * H) / OH, h1 = ((oh + 1) * H + OH - 1) / OH;
    const int w0 = (ow * W) / OW, w1 = ((ow + 1) * W + OW - 1) / OW;
    const float* xb = x + (size_t)b * H * W * C + (cq << 2);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = h0; h < h1; ++h)
        for (int w_ = w0; w_ < w1; ++w_) {
            const float4 v = *reinterpret_cast<const float4*>(xb + ((size_t)h * W + w_) * C);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
    s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
    *reinterpret_cast<float4*>(y + (((size_t)b * OH + oh) * OW + ow) * C + (cq << 2)) = s;
}

// bf16 input, fp32 output (the last activation of a bf16-stored encoder becomes the fp32 feature vector here)
__global__ void adaptive_avgpool_bf16_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, int H, int W, int C, int OH,
                                             int OW, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over B*OH*OW*(C/4)
    if (i >= total) return;
    const int nq = C >> 2;
    const int cq = (int)(i % nq);
    long long pix = i / nq;
    const int ow = (int)(pix % OW); pix /= OW;
    const int oh = (int)(pix % OH);
    const long long b = pix / OH;
    const int h0 = (oh * H) / OH, h1 = ((oh + 1) * H + OH - 1) / OH;
    const int w0 = (ow * W) / OW, w1 = ((ow + 1) * W + OW - 1) / OW;
    const unsigned short* xb = x + (size_t)b * H * W * C + (cq << 2);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = h0; h < h1; ++h)
        for (int w_ = w0; w_ < w1; ++w_) {
            const uint2 v = *reinterpret_cast<const uint2*>(xb + ((size_t)h * W + w_) * C);
            s.x += __uint_as_float(v.x << 16); s.y += __uint_as_float(v.x & 0xffff0000u);
            s.z += __uint_as_float(v.y << 16); s.w += __uint_as_float(v.y & 0xffff0000u);
        }
    const float inv = 1.0f / (float)((h1 - h0) * (w1 - w0));
    s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
    *reinterpret_cast<float4*>(y + (((size_t)b * OH + oh) * OW + ow) * C + (cq << 2)) = s;
}

int adaptive_avgpool2d_bf16_f32(const unsigned short* x, float* y, int B, int H, int W, int C, int OH, int OW, hipStream_t st) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) {
        set_error("adaptive_avgpool2d_bf16: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (C % 4 || !aligned16(x) || !aligned16(y)) { set_error("adaptive_avgpool2d_bf16: C %% 4 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    const long long total = (long long)B * OH * OW * (C / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) { set_error("adaptive_avgpool2d_bf16: too large"); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(adaptive_avgpool_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, H, W, C, OH, OW, total);
    return launch_status("adaptive_avgpool2d_bf16");
}

int adaptive_avgpool2d_f32(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, hipStream_t st) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0) {
        set_error("adaptive_avgpool2d: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (C % 4 || !aligned16(x) || !aligned16(y)) { set_error("adaptive_avgpool2d: C %% 4 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    const long long total = (long long)B * OH * OW * (C / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) { set_error("adaptive_avgpool2d: too large"); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, H, W, C, OH, OW, total);
    return launch_status("adaptive_avgpool2d");
}

}  // namespace icap
