// MBConv front half for the low-resolution blocks of EfficientNet-B0 (14 x 14 and 7 x 7, stride 1), whole images per
// workgroup:  y = swish(bn(depthwise(swish(bn(x . We^T)))))  plus the per-image channel sums squeeze-excite needs.
//
// At these resolutions the 1x1 expand convolution is a GEMM with a 480-1152 wide result that the depthwise convolution
// reads straight back: 0.4-0.5 GB written and re-read per layer at 1024 images.  A depthwise 3x3 / 5x5 at 14 x 14 needs
// no neighbours from outside the image, so a workgroup that owns WHOLE images and a 64-channel slice computes
//   1. E[196 pixels][64 ch] = x[196][Cin] . We[64][Cin]^T on v_mfma_f32_32x32x2_f32 (A and We staged through LDS in
//      K slices of 16, double-buffered; 196 rows = one 14 x 14 image or four 7 x 7 images, padded to 8 row blocks of
//      which the last is skipped), BatchNorm + swish applied to the accumulators, E written to LDS;
//   2. the depthwise convolution from LDS: a thread owns one image row and four channels, keeps the row's outputs in
//      registers and walks the filter rows (column bounds are compile-time, row bounds one test per filter row);
//      BatchNorm + swish, 16-byte stores, per-image channel sums in a fixed order;
// and the expanded tensor never exists in HBM.  Workgroups of one image run on the same XCD back to back (the image's
// 63-150 KB of input is fetched from HBM once and re-read from that L2 by the other channel slices).
#include "common.hpp"

namespace icap {

typedef float f32x16l __attribute__((ext_vector_type(16)));

constexpr int XL_ROWS = 196;          // pixels per workgroup: one 14 x 14 image or four 7 x 7 images
constexpr int XL_CH = 64;             // expanded channels per workgroup
constexpr int XL_BK = 16;             // contraction slice
constexpr int XL_LDK = XL_BK + 4;     // padded staging row (80 bytes)
constexpr int XL_ES = XL_CH + 4;      // padded row of E

__device__ __forceinline__ float4 xl_keep(bool ok, float4 v) {   // by value: a select between named objects pins them to memory
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

template <int CIN, int KS, int HW>
__global__ __launch_bounds__(256, 2) void expand_dw_img_kernel(const float* __restrict__ x, const float* __restrict__ we,
                                                               const float* __restrict__ es, const float* __restrict__ eb,
                                                               const float* __restrict__ wd, const float* __restrict__ ds,
                                                               const float* __restrict__ db, float* __restrict__ y,
                                                               float* __restrict__ pooled, int B, int Cexp, int nchunk) {
    static_assert(HW == 14 || HW == 7, "whole-image tiles: 14 x 14 (one image) or 7 x 7 (four images)");
    static_assert(CIN % XL_BK == 0 && (KS == 3 || KS == 5), "contraction in slices of 16; 3x3 or 5x5 filters");
    constexpr int IMGS = HW == 14 ? 1 : 4, PX = HW * HW, NK = CIN / XL_BK, PAD = KS / 2, NT = IMGS * HW;
    constexpr int STAGE_FLOATS = 2 * (256 + XL_CH) * XL_LDK, E_FLOATS = XL_ROWS * XL_ES;
    __shared__ __attribute__((aligned(16))) float smem[STAGE_FLOATS > E_FLOATS ? STAGE_FLOATS : E_FLOATS];
    __shared__ float red[NT][XL_CH];
    __shared__ __attribute__((aligned(16))) float wds[KS * KS][XL_CH];
    float* As = smem;                               // [2][256][XL_LDK]
    float* Bs = smem + 2 * 256 * XL_LDK;            // [2][XL_CH][XL_LDK]
    float* E = smem;                                // [XL_ROWS][XL_ES], after the K loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    // XCD x walks a contiguous range of the (image group, channel slice) list, slice fastest
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    const int t_lin = xcd * (nwg >> 3) + (xcd < (nwg & 7) ? xcd : (nwg & 7)) + slot0;
    const int grp = t_lin / nchunk, chunk = t_lin - grp * nchunk;
    const int img0 = grp * IMGS, ch0 = chunk * XL_CH;

    // ---- staging coordinates: four lanes cover a row's 64 bytes; the two rows one group of 8 lanes writes are 4 rows apart
    // (rows are 80 bytes: 4 x 20 dwords = 16 banks), so an LDS store instruction meets no bank twice
    const int k4 = (tid & 3) * 4, q8 = tid >> 2;
    const int srow = (q8 & 1) * 4 + ((q8 >> 1) & 3) + (q8 >> 3) * 8;          // 0..63
    const float* a_src[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + 64 * i;
        a_ok[i] = row < XL_ROWS && img0 + row / PX < B;
        a_src[i] = x + ((size_t)img0 * PX + (a_ok[i] ? row : 0)) * CIN + k4;
    }
    const bool b_ok = ch0 + srow < Cexp;
    const float* b_src = we + (size_t)(b_ok ? ch0 + srow : 0) * CIN + k4;
    // Two register sets: the loads of step t + 3 are issued when step t + 1 has gone to LDS, so a load has two steps of
    // matrix-core work to arrive (with one set the K loop, a handful of short steps, runs at one L2 latency per step).
    // The loop is unrolled completely (NK <= 12), which makes the set of every step a compile-time choice.
    float4 ra[2][4], rb[2];
    auto gload = [&](int t, int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[set][i] = *reinterpret_cast<const float4*>(a_src[i] + t * XL_BK);
        rb[set] = *reinterpret_cast<const float4*>(b_src + t * XL_BK);
    };
    auto lds_store = [&](int buf, int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(&As[(buf * 256 + srow + 64 * i) * XL_LDK + k4]) = xl_keep(a_ok[i], ra[set][i]);
        *reinterpret_cast<float4*>(&Bs[(buf * XL_CH + srow) * XL_LDK + k4]) = xl_keep(b_ok, rb[set]);
    };

    // the depthwise filter slice of this workgroup goes to LDS now (read in the second phase, a global-memory latency away
    // from every filter row otherwise)
    for (int idx = tid; idx < KS * KS * (XL_CH / 4); idx += 256) {
        const int tap = idx / (XL_CH / 4), c4 = (idx - tap * (XL_CH / 4)) * 4;
        const bool ok = ch0 + c4 < Cexp;
        const float4 v = *reinterpret_cast<const float4*>(wd + (size_t)tap * Cexp + (ok ? ch0 + c4 : 0));
        *reinterpret_cast<float4*>(&wds[tap][c4]) = xl_keep(ok, v);
    }

    // ---- expand GEMM: wave w owns row blocks 2w and 2w + 1 (the eighth block holds no pixel) and both column blocks
    f32x16l acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nrb = wave == 3 ? 1 : 2;
    gload(0, 0);
    if (NK > 1) gload(1, 1);
    lds_store(0, 0);
    if (NK > 2) gload(2, 0);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NK; ++t) {
        const int cur = t & 1;
        float af[2][8], bf[2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* src = &Bs[(cur * XL_CH + 32 * j + lr) * XL_LDK + 8 * lh];
            const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
            bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w;
            bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < nrb) {
                const float* src = &As[(cur * 256 + (2 * wave + i) * 32 + lr) * XL_LDK + 8 * lh];
                const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
                af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }
        }
        if (t + 1 < NK) {
            lds_store(cur ^ 1, (t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);         // the new loads stay behind the stores that free their registers
            if (t + 3 < NK) gload(t + 3, (t + 1) & 1);
        }
        __syncthreads();
    }

    // ---- E = swish(bn(acc)) into LDS (the staging buffers are dead: every wave passed the loop's last barrier)
    {
        float s_[2], b_[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = ch0 + 32 * j + lr;
            const bool ok = c < Cexp;
            s_[j] = (ok && es) ? es[c] : 1.f;
            b_[j] = (ok && eb) ? eb[c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < nrb) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = (2 * wave + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        const float v = fmaf(acc[i][j][e], s_[j], b_[j]);
                        if (row < XL_ROWS) E[row * XL_ES + 32 * j + lr] = v * sigmoid_fast(v);
                    }
            }
        }
    }
    __syncthreads();

    // ---- depthwise convolution from E: thread = (channel quad, row slot); a task is one image row
    const int q = tid & 15, rslot = tid >> 4;
    const int cb = ch0 + q * 4;
    const bool c_ok = cb < Cexp;
    const int cbc = c_ok ? cb : 0;
    const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f), zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 dsv = ds ? *reinterpret_cast<const float4*>(ds + cbc) : one4;
    const float4 dbv = db ? *reinterpret_cast<const float4*>(db + cbc) : zero4;
#pragma unroll 1
    for (int task = rslot; task < NT; task += 16) {
        const int img = task / HW, r = task - img * HW;
        float4 o[HW];
#pragma unroll
        for (int c = 0; c < HW; ++c) o[c] = zero4;
#pragma unroll
        for (int dr = 0; dr < KS; ++dr) {
            const int rr = r + dr - PAD;
            if (rr < 0 || rr >= HW) continue;
            float4 w[KS];
#pragma unroll
            for (int dc = 0; dc < KS; ++dc) w[dc] = *reinterpret_cast<const float4*>(&wds[dr * KS + dc][q * 4]);
            const float* erow = &E[(img * PX + rr * HW) * XL_ES + q * 4];
            float4 ev[HW];
#pragma unroll
            for (int c = 0; c < HW; ++c) ev[c] = *reinterpret_cast<const float4*>(erow + c * XL_ES);
#pragma unroll
            for (int c = 0; c < HW; ++c)
#pragma unroll
                for (int dc = 0; dc < KS; ++dc) {
                    const int cc = c + dc - PAD;
                    if (cc >= 0 && cc < HW) {
                        o[c].x = fmaf(ev[cc].x, w[dc].x, o[c].x);      // explicit: the library is built without contraction
                        o[c].y = fmaf(ev[cc].y, w[dc].y, o[c].y);
                        o[c].z = fmaf(ev[cc].z, w[dc].z, o[c].z);
                        o[c].w = fmaf(ev[cc].w, w[dc].w, o[c].w);
                    }
                }
        }
        float4 ps = zero4;
        const bool st_ok = c_ok && img0 + img < B;
        float* dst = y + (((size_t)(img0 + img) * PX + r * HW) * Cexp + cbc);
#pragma unroll
        for (int c = 0; c < HW; ++c) {
            float4 v;
            v.x = fmaf(o[c].x, dsv.x, dbv.x); v.y = fmaf(o[c].y, dsv.y, dbv.y); v.z = fmaf(o[c].z, dsv.z, dbv.z); v.w = fmaf(o[c].w, dsv.w, dbv.w);
            v.x *= sigmoid_fast(v.x); v.y *= sigmoid_fast(v.y); v.z *= sigmoid_fast(v.z); v.w *= sigmoid_fast(v.w);
            if (st_ok) *reinterpret_cast<float4*>(dst + (size_t)c * Cexp) = v;
            ps.x += v.x; ps.y += v.y; ps.z += v.z; ps.w += v.w;
        }
        *reinterpret_cast<float4*>(&red[task][q * 4]) = ps;
    }
    if (!pooled) return;
    __syncthreads();
    // per-image channel sums: the image's rows in order
    for (int idx = tid; idx < IMGS * XL_CH; idx += 256) {
        const int img = idx / XL_CH, c = idx - img * XL_CH;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < HW; ++r) s += red[img * HW + r][c];
        if (ch0 + c < Cexp && img0 + img < B) pooled[(size_t)(img0 + img) * Cexp + ch0 + c] = s;
    }
}

// 1 when this layer shape runs on the whole-image kernel (stride 1, 'same' padding, 14 x 14 with 80 / 112 input channels or
// 7 x 7 with 192)
int expand_dwconv_whole_image(int H, int W, int Cin, int Cexp, int KS, int stride) {
    if (stride != 1 || H != W || (KS != 3 && KS != 5) || Cexp <= 0 || Cexp % 4) return 0;
    return (H == 14 && (Cin == 80 || Cin == 112)) || (H == 7 && Cin == 192);
}

int expand_dwconv_img_f32(const float* x, const float* we, const float* es, const float* eb, const float* wd, const float* ds,
                          const float* db, float* y, float* pooled, int B, int H, int W, int Cin, int Cexp, int KS, int pad_t,
                          int pad_l, hipStream_t st) {
    if (!expand_dwconv_whole_image(H, W, Cin, Cexp, KS, 1) || pad_t != KS / 2 || pad_l != KS / 2 || B <= 0) {
        set_error("expand_dwconv (whole image): unsupported layer %dx%d Cin=%d %dx%d pad %d,%d", H, W, Cin, KS, KS, pad_t, pad_l);
        return ICAP_ERR_ARG;
    }
    const int imgs = H == 14 ? 1 : 4;
    const int nchunk = (Cexp + XL_CH - 1) / XL_CH;
    const long long wgs = (long long)((B + imgs - 1) / imgs) * nchunk;
    if (wgs > 0x7fffffffLL) { set_error("expand_dwconv (whole image): problem too large"); return ICAP_ERR_LIMIT; }
    const dim3 grid((unsigned)wgs), block(256);
#define ICAP_XL(CIN_, KS_, HW_) \
    hipLaunchKernelGGL((expand_dw_img_kernel<CIN_, KS_, HW_>), grid, block, 0, st, x, we, es, eb, wd, ds, db, y, pooled, B, Cexp, nchunk)
    if (Cin == 80 && KS == 3) ICAP_XL(80, 3, 14);
    else if (Cin == 80) ICAP_XL(80, 5, 14);
    else if (Cin == 112 && KS == 3) ICAP_XL(112, 3, 14);
    else if (Cin == 112) ICAP_XL(112, 5, 14);
    else if (KS == 3) ICAP_XL(192, 3, 7);
    else ICAP_XL(192, 5, 7);
#undef ICAP_XL
    return launch_status("expand_dwconv (whole image)");
}

}  // namespace icap
