// Fused MBConv front half: 1x1 expand convolution + BatchNorm + swish + depthwise KSxKS convolution +
// BatchNorm + swish (+ the per-channel sums squeeze-excite needs), without the expanded tensor ever
// reaching HBM.  In EfficientNet-B0's early blocks that tensor is 6x the block's input and its write +
// re-read dominates the time of the two separate kernels.
//
// A workgroup (4 waves) owns one TH x TW tile of output pixels of one image.  The input pixels the tile's
// depthwise windows cover (IH x IW, with halo) are the rows of a small GEMM against the expand weights:
//   * each wave keeps the A fragments of its 32-pixel row blocks in registers for the whole kernel
//     (v_mfma_f32_32x32x2_f32: lane (r, h) supplies pixel r, input channels h*CIN/2 ...),
//   * per 32-channel chunk: B fragments straight from the (L2-resident) weights, CIN/2 MFMAs per row block,
//     BatchNorm + swish on the accumulators, zero for pixels outside the image (the depthwise convolution
//     pads the EXPANDED activations with zeros), result to LDS as E[pixel][32 channels];
//   * depthwise phase: a thread owns a channel quad and a strip of SW output pixels, walks the filter rows
//     over E with 16-byte LDS reads, applies BatchNorm + swish and writes 16 B per lane along the channel axis.
// Several workgroups share a CU (E is 40-60 KB), so one workgroup's MFMA phase overlaps another's depthwise
// phase and the global stores.
#include "common.hpp"

namespace icap {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 xd_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xd_bf16x2 __attribute__((ext_vector_type(2)));
typedef float xd_f32x2 __attribute__((ext_vector_type(2)));
typedef float xd_f32x4 __attribute__((ext_vector_type(4)));
typedef short xd_s16x4 __attribute__((ext_vector_type(4)));

constexpr int XD_CC = 32;            // expanded channels per chunk (one MFMA column block)
constexpr int XD_LDE = XD_CC + 4;    // padded row of E (16-byte aligned rows, consecutive pixels start 4 banks apart)

// x * sigmoid(x) on a pair: the multiplies and the add as packed fp32 instructions, exp2 and rcp per element
__device__ __forceinline__ xd_f32x2 xd_swish2(xd_f32x2 v) {
    const xd_f32x2 nk2 = {-1.4426950408889634f, -1.4426950408889634f}, one2 = {1.f, 1.f};
    xd_f32x2 t = v * nk2;
    t.x = __builtin_amdgcn_exp2f(t.x); t.y = __builtin_amdgcn_exp2f(t.y);
    t = t + one2;
    t.x = __builtin_amdgcn_rcpf(t.x); t.y = __builtin_amdgcn_rcpf(t.y);
    return v * t;
}

// Workgroups whose LDS footprint fits a CU three times are compiled for three waves per SIMD, the others for two (and so is the
// 40-channel stride-2 layer on split products: at three it spills, and a spill reload between two output stores, or before a
// barrier, waits for the stores themselves - measured 2 % slower than two waves without spills).
template <int KS, int S, int TH, int TW, bool WIDE = false>
constexpr int xd_waves() {
    if (WIDE) return 2;   // the stem's bf16 variant holds the fp32 patch and its packed copy at once: 256-register budget
    return ((((TH - 1) * S + KS) * ((TW - 1) * S + KS) + 31) / 32) * 32 * XD_LDE * 4 + 2 * (KS * KS + 1) * 128 + 512 <= 160 * 1024 / 3 ? 3 : 2;
}

// STEM: the first convolution is not a 1x1 expand but EfficientNet's 3x3 stride-2 stem on the 4-channel-padded image
// (x [B][img_h][img_w][4], we [Cexp][3][3][4]): the GEMM's K axis is the 36-element patch, padded to 40 (ten taps, five per
// lane half, the tenth zero), gathered straight from the image; H x W is then the stem's OUTPUT size.
// PROD: how the expand GEMM multiplies.  0: v_mfma_f32_32x32x2_f32.  1: operands rounded to bf16, v_mfma_f32_32x32x16_bf16.
// 3: fp32-accurate on the bf16 instruction - both operands as exact three-way bf16 splits (hi, mid, lo), the six products
// hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi per K block (the dropped three are <= 2^-24 of the product): for CIN = 24,
// 12 bf16 MFMAs of 32 cycles instead of 12 fp32 MFMAs of 64.
// ALLC > 0: the layer has at most ALLC channel chunks and the workgroup keeps EVERY chunk's parameters on chip from the start
// (packed B planes and biases in registers, depthwise rows in LDS): no load inside the chunk loop, so no wait that would also
// wait for the previous chunk's output stores (vmcnt counts loads and stores in one order).  Only where registers and LDS
// allow it without losing a workgroup per CU: the 16-channel layer (three chunks).
template <int CIN, int KS, int S, int TH, int TW, int SW, bool STEM = false, int PROD = 0, int ALLC = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ALLC ? 4 : (PROD == 3 && S == 2 && CIN == 40) ? 2 : xd_waves<KS, S, TH, TW, STEM && PROD == 1>()))) void expand_dw_kernel(
    const float* __restrict__ x, const float* __restrict__ we, const float* __restrict__ es, const float* __restrict__ eb,
    const float* __restrict__ wd, const float* __restrict__ dsc, const float* __restrict__ dbi, float* __restrict__ y,
    float* __restrict__ partial, int H, int W, int Cexp, int pad_t, int pad_l, int Ho, int Wo, int tiles_w, int cpg,
    int img_h, int img_w, int stem_pad_t, int stem_pad_l, int flags) {
    const int out16 = flags & 1;            // flags: bit 0 = bf16 output tensor, bit 1 (stem) = x is the planar [B][3][img_h][img_w] image
    static_assert(!STEM || CIN == 40, "the stem patch is 36 values padded to 40");
    static_assert(PROD == 0 || PROD == 1 || (PROD == 3 && !STEM), "split products: expanding blocks only");
    static_assert(ALLC == 0 || (PROD == 3 && ALLC <= 3), "all-chunk parameters: split products, at most three chunks");
    constexpr bool BF16 = PROD != 0;       // operands go through packed bf16 registers
    constexpr int NPL = PROD == 3 ? 3 : 1;  // bf16 planes per operand
    constexpr int IH = (TH - 1) * S + KS, IW = (TW - 1) * S + KS, P = IH * IW;
    constexpr int MB = (P + 31) / 32;          // 32-pixel row blocks
    constexpr int NRB = (MB + 3) / 4;          // row blocks per wave
    constexpr int KH2 = CIN / 2;               // input channels per lane half
    constexpr int SPR = (TW + SW - 1) / SW;    // strips per output row
    constexpr int NSTRIP = TH * SPR;
    constexpr int NIN = (SW - 1) * S + KS;     // E columns feeding one strip
    static_assert(CIN % 8 == 0, "each lane half reads whole float4s");

    __shared__ __attribute__((aligned(16))) float E[MB * 32 * XD_LDE];
    __shared__ float4 red[4][8];
    // depthwise weights times the folded-BN scale (KS*KS rows) and the folded-BN bias (1 row) of a 32-channel chunk,
    // double buffered: the next chunk's rows are fetched into a register at the top of a chunk and stored behind its
    // MFMA phase
    constexpr int WROWS = KS * KS + 1;
    static_assert(WROWS * 8 <= 256, "one float4 of the chunk's depthwise parameters per thread");
    // ALLC: one block per chunk, without the bias row (kept in the pad columns of E)
    __shared__ float4 Wl[ALLC ? ALLC : 2][ALLC ? KS * KS : WROWS][8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y;
    const int tile = blockIdx.x;
    const int th_i = tile / tiles_w;
    const int ho0 = th_i * TH, wo0 = (tile - th_i * tiles_w) * TW;
    const int hi0 = ho0 * S - pad_t, wi0 = wo0 * S - pad_l;

    // ---- A fragments: this wave's pixels, all CIN input channels split over the two lane halves ----
    // Nothing here consumes a loaded value (a consumer would wait for the load where it is issued): pixels outside the
    // image read a valid dummy address and their rows are zeroed after the MFMAs; only the stem's image-border taps need
    // true zeros, applied after the first chunk's parameter loads have been issued (tapmask bit j: tap j of this lane is real).
    float af[NRB][KH2];
    unsigned rowmask[NRB], tapmask[NRB];
#pragma unroll
    for (int i = 0; i < NRB; ++i) {
        const int pi = (wave + 4 * i) * 32 + lr;
        const int ph = pi / IW, pw = pi - ph * IW;
        const int hi = hi0 + ph, wi = wi0 + pw;
        const bool ok = pi < P && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        tapmask[i] = 0x1f;
        if constexpr (STEM) {
            tapmask[i] = 0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int tap = lh * 5 + j;                     // tap 9 does not exist: its weights are zero
                const int kh = tap / 3, kw = tap - kh * 3;
                const int ih = hi * 2 - stem_pad_t + kh, iw = wi * 2 - stem_pad_l + kw;
                const bool tok = ok && tap < 9 && (unsigned)ih < (unsigned)img_h && (unsigned)iw < (unsigned)img_w;
                if (flags & 2) {    // the reference's NCHW tensor as it is: one value per colour plane, the fourth channel is zero
                    const size_t plane = (size_t)img_h * img_w;
                    const float* src = tok ? x + ((size_t)b * 3 * img_h + ih) * img_w + iw : x;
                    af[i][4 * j] = src[0]; af[i][4 * j + 1] = src[tok ? plane : 0]; af[i][4 * j + 2] = src[tok ? 2 * plane : 0]; af[i][4 * j + 3] = 0.f;
                } else {
                    const float4 v = *reinterpret_cast<const float4*>(tok ? x + (((size_t)b * img_h + ih) * img_w + iw) * 4 : x);
                    af[i][4 * j] = v.x; af[i][4 * j + 1] = v.y; af[i][4 * j + 2] = v.z; af[i][4 * j + 3] = v.w;
                }
                tapmask[i] |= (tok || !ok || tap >= 9) ? (1u << j) : 0u;
            }
        } else {
            const float* src = x + (size_t)b * H * W * CIN + (ok ? (unsigned)((hi * W + wi) * CIN + lh * KH2) : 0u);
#pragma unroll
            for (int k = 0; k < KH2; k += 4) {
                const float4 v = *reinterpret_cast<const float4*>(src + k);
                af[i][k] = v.x; af[i][k + 1] = v.y; af[i][k + 2] = v.z; af[i][k + 3] = v.w;
            }
        }
        // bit r set: row r needs no zeroing (rows past the tile's last pixel are never read by the depthwise phase);
        // lanes 0..31 are the lh = 0 copy of rows 0..31
        rowmask[i] = (unsigned)__ballot(ok || pi >= P);
    }

    const int nchunks = (Cexp + XD_CC - 1) / XD_CC;
    const int ch_end = min(nchunks, ((int)blockIdx.z + 1) * cpg);
    const int q = tid & 7, sidx = tid >> 3;
    // per-image bases are wave-uniform (scalar registers); what a thread adds fits 32 bits (the host checks the image size)
    const size_t ybase = (size_t)b * Ho * Wo * Cexp;

    // Per-chunk parameters are fetched one chunk ahead so that no phase waits for a global load it has just issued; the
    // fetched values are consumed (scaled, stored to LDS) only when the chunk that fetched them hands over.
    // B fragment, raw: fold_b() multiplies the folded-BN scale into it (a lane's B values all belong to its one output
    // channel); the bias seeds the accumulator, so the MFMA chain ends with the normalised value.
    float bf[KH2], bi;
    auto load_expand_params = [&](int ch, float (&f)[KH2], float& s_, float& b_) {
        const int cn = ch * XD_CC + lr;
        const bool cok = cn < Cexp;
        constexpr int WLD = STEM ? 36 : CIN;      // floats per weight row in memory
        const float* wsrc = we + (unsigned)((cok ? cn : 0) * WLD + lh * KH2);
        s_ = (cok && es) ? es[cn] : (cok ? 1.f : 0.f);
        b_ = (cok && eb) ? eb[cn] : 0.f;
#pragma unroll
        for (int k = 0; k < KH2; k += 4) {
            const bool kok = !STEM || lh * KH2 + k < 36;
            const float4 v = *reinterpret_cast<const float4*>(kok ? wsrc + k : we);
            f[k] = v.x; f[k + 1] = v.y; f[k + 2] = v.z; f[k + 3] = v.w;
        }
    };
    auto fold_b = [&](float (&dst)[KH2], const float (&f)[KH2], float s_) {
#pragma unroll
        for (int k = 0; k < KH2; ++k) dst[k] = (!STEM || lh * KH2 + (k & ~3) < 36) ? f[k] * s_ : 0.f;
    };
    // thread t < WROWS*8 owns float4 (row t/8, quad t%8) of the chunk's depthwise parameter block: a weight row (to be
    // multiplied by the folded-BN scale, fold_dw) or the bias row
    const int wrow = tid >> 3;
    // Every thread issues the same two loads from valid addresses, whatever its row means; what the values mean (weight row
    // x scale, bias row, nothing) is sorted out where they are used (fold_dw).  Loads under divergent branches would make the
    // compiler wait for them at the joins - right where they were issued, in every chunk.
    const bool dw_wt = wrow < KS * KS;                 // this thread's row is a filter tap (else: the bias row, or none)
    const float* dw_vsrc = dw_wt ? wd + (size_t)wrow * Cexp : (dbi ? dbi : wd);
    const float* dw_ssrc = dsc ? dsc : wd;
    auto load_dw_params = [&](int ch, float4& v, float4& s4) {
        const int c = ch * XD_CC + q * 4;
        const int cc = c < Cexp ? c : 0;
        v = *reinterpret_cast<const float4*>(dw_vsrc + cc);
        s4 = *reinterpret_cast<const float4*>(dw_ssrc + cc);
    };
    auto fold_dw = [&](int ch, const float4& v, const float4& s4) {
        const bool live = wrow < WROWS && ch * XD_CC + q * 4 < Cexp && (dw_wt || dbi);
        const bool scaled = dw_wt && dsc;
        const float4 s = scaled ? s4 : make_float4(1.f, 1.f, 1.f, 1.f);
        return live ? make_float4(v.x * s.x, v.y * s.y, v.z * s.z, v.w * s.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // bf16 products (BF16): a lane's KH2 values become ceil(KH2 / 8) packed chunks of 8 (zero padded); chunk c of both
    // lane halves is one K = 16 block of v_mfma_f32_32x32x16_bf16 - A and B use the same assignment, so any works
    constexpr int NCH = (KH2 + 7) / 8;
    auto pack8 = [&](const float (&f)[KH2], uint4 (&o)[NCH]) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 8 * c + 2 * j;
                const xd_f32x2 v = {k < KH2 ? f[k < KH2 ? k : 0] : 0.f, k + 1 < KH2 ? f[k + 1 < KH2 ? k + 1 : 0] : 0.f};
                w[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, xd_bf16x2));
            }
            o[c] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    };
    // residual of f after its packed bf16 heads o (exact in fp32): the next plane of a split is the head of this
    auto residual8 = [&](float (&f)[KH2], const uint4 (&o)[NCH]) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const unsigned w[4] = {o[c].x, o[c].y, o[c].z, o[c].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 8 * c + 2 * j;
                if (k < KH2) f[k] -= __uint_as_float(w[j] << 16);
                if (k + 1 < KH2) f[k + 1] -= __uint_as_float(w[j] & 0xffff0000u);
            }
        }
    };
    auto pack_planes = [&](const float (&f)[KH2], uint4 (&o)[NPL][NCH]) {
        pack8(f, o[0]);
        if constexpr (NPL == 3) {
            float r[KH2];
#pragma unroll
            for (int k = 0; k < KH2; ++k) r[k] = f[k];
            residual8(r, o[0]);
            pack8(r, o[1]);
            residual8(r, o[1]);
            pack8(r, o[2]);
        }
    };
    uint4 a16[BF16 ? NRB : 1][NPL][NCH], b16[NPL][NCH];
    const int ch_begin = blockIdx.z * cpg;
    uint4 b16all[ALLC > 1 ? ALLC - 1 : 1][NPL][NCH];     // chunks 1 .. ALLC - 1 (chunk 0's planes go straight into b16)
    float biall[ALLC > 1 ? ALLC - 1 : 1];
    if constexpr (ALLC > 0) {
        // every chunk's parameters, once (the host launches this form only for <= ALLC chunks in one z slice)
#pragma unroll
        for (int c = 0; c < ALLC; ++c) {
            const int chc = c < ch_end ? c : ch_end - 1;
            float braw[KH2], s_;
            load_expand_params(chc, braw, s_, c == 0 ? bi : biall[c ? c - 1 : 0]);
            float4 wv, ws4;
            load_dw_params(chc, wv, ws4);
            fold_b(bf, braw, s_);
            if (c == 0) pack_planes(bf, b16);
            else pack_planes(bf, b16all[c ? c - 1 : 0]);
            if (wrow < KS * KS) Wl[c][wrow][q] = fold_dw(chc, wv, ws4);
            // the bias row goes into the four pad floats of E's rows c * 8 + q (columns 32..35: nothing else writes them)
            else if (wrow == KS * KS) *reinterpret_cast<float4*>(&E[(c * 8 + q) * XD_LDE + XD_CC]) = fold_dw(chc, wv, ws4);
        }
    } else {
        float braw[KH2], s_;
        load_expand_params(ch_begin, braw, s_, bi);
        float4 wv, ws4;
        load_dw_params(ch_begin, wv, ws4);
        fold_b(bf, braw, s_);
        if constexpr (BF16) pack_planes(bf, b16);
        if (wrow < WROWS) Wl[ch_begin & 1][wrow][q] = fold_dw(ch_begin, wv, ws4);
    }
    // The A fragments take their final form once, here (after the first chunk's parameter loads have been issued): the
    // stem's border taps become zeros, and for the bf16 products the pixels are packed (and split) so that the chunk loop
    // holds only the packed planes.
#pragma unroll
    for (int i = 0; i < NRB; ++i) {
        if (wave + 4 * i < MB) {
            if constexpr (STEM) {
                if (__any(tapmask[i] != 0x1f)) {
#pragma unroll
                    for (int j = 0; j < 5; ++j)
                        if (!((tapmask[i] >> j) & 1u)) { af[i][4 * j] = 0.f; af[i][4 * j + 1] = 0.f; af[i][4 * j + 2] = 0.f; af[i][4 * j + 3] = 0.f; }
                }
            }
            if constexpr (BF16) pack_planes(af[i], a16[i]);
        }
    }

    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int c0 = ch * XD_CC;
        const int chn = ch + 1 < ch_end ? ch + 1 : ch;
        float bfn[KH2], scn, bin;
        float4 wln, wsn;
        if constexpr (ALLC == 0) {
            load_expand_params(chn, bfn, scn, bin);
            load_dw_params(chn, wln, wsn);
        }
        // ---- expand: E = swish(bn(A W^T)) for 32 channels ----
        {
#pragma unroll
            for (int i = 0; i < NRB; ++i) {
                const int rb = wave + 4 * i;
                if (rb < MB) {
                    // the accumulator starts at zero (no register initialisation: the first MFMA takes the constant) and
                    // the folded-BN bias joins in the packed epilogue
                    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if constexpr (BF16) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            // the last chunk of a lane half that holds 4 (not 8) values is a K = 8 block: the halves of its
                            // packed words that would only carry zeros are never read, so they cost no registers
                            auto mm = [&](const uint4& a, const uint4& b) {
                                if constexpr (KH2 % 8 == 4) {
                                    if (c == NCH - 1) {
                                        acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(xd_s16x4, make_uint2(a.x, a.y)),
                                                                                       __builtin_bit_cast(xd_s16x4, make_uint2(b.x, b.y)), acc, 0, 0, 0);
                                        return;
                                    }
                                }
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(xd_bf16x8, a), __builtin_bit_cast(xd_bf16x8, b), acc, 0, 0, 0);
                            };
                            if constexpr (NPL == 3) {   // smallest terms first: (a plane, b plane) with plane sum <= 2
                                constexpr int pa[5] = {2, 0, 1, 1, 0}, pb[5] = {0, 2, 1, 0, 1};
#pragma unroll
                                for (int t5 = 0; t5 < 5; ++t5) mm(a16[i][pa[t5]][c], b16[pb[t5]][c]);
                            }
                            mm(a16[i][0][c], b16[0][c]);
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < KH2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[s], acc, 0, 0, 0);
                    }
                    // bias + swish on accumulator pairs (packed fp32 instructions), stage by stage over all eight pairs so
                    // that no exp2 / rcp result is needed right after it is issued
                    const xd_f32x2 bi2 = {bi, bi}, nk2 = {-1.4426950408889634f, -1.4426950408889634f}, one2 = {1.f, 1.f};
                    float r[16];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {   // eight accumulators at a time: four live temporaries
                        xd_f32x2 z[4], t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const xd_f32x2 a2 = {acc[8 * hf + 2 * e], acc[8 * hf + 2 * e + 1]}; z[e] = a2 + bi2; }
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = z[e] * nk2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { t[e].x = __builtin_amdgcn_exp2f(t[e].x); t[e].y = __builtin_amdgcn_exp2f(t[e].y); }
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = t[e] + one2;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { t[e].x = __builtin_amdgcn_rcpf(t[e].x); t[e].y = __builtin_amdgcn_rcpf(t[e].y); }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const xd_f32x2 o2 = z[e] * t[e]; r[8 * hf + 2 * e] = o2.x; r[8 * hf + 2 * e + 1] = o2.y; }
                    }
                    float* dst = &E[(rb * 32 + 4 * lh) * XD_LDE + lr];
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[((e & 3) + 8 * (e >> 2)) * XD_LDE] = r[e];
                    // rows of pixels outside the image become zeros (a wave-uniform test: border tiles only): lane (lr, lh)
                    // clears half lh of row lr with 16-byte writes; a wave's LDS writes keep their order
                    if (rowmask[i] != 0xffffffffu) {
                        if (!((rowmask[i] >> lr) & 1u)) {
                            float4* zrow = reinterpret_cast<float4*>(&E[(rb * 32 + lr) * XD_LDE + lh * 16]);
#pragma unroll
                            for (int j = 0; j < 4; ++j) zrow[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                }
            }
        }
        if constexpr (ALLC > 0) {
            // the next chunk's planes and bias out of the registers that hold every chunk's (wave-uniform selection)
#pragma unroll
            for (int c = 1; c < ALLC; ++c)
                if (chn == c) {
#pragma unroll
                    for (int p = 0; p < NPL; ++p)
#pragma unroll
                        for (int k = 0; k < NCH; ++k) b16[p][k] = b16all[c - 1][p][k];
                    bi = biall[c - 1];
                }
        } else {
            if (wrow < WROWS) Wl[(ch + 1) & 1][wrow][q] = fold_dw(chn, wln, wsn);
            // the next chunk's B fragments take their place now, not at the end of the chunk: a wait for their loads there
            // would also wait for the depthwise phase's stores, which have only just been issued
            fold_b(bf, bfn, scn);
            if constexpr (BF16) pack_planes(bf, b16);
            bi = bin;
        }
        __syncthreads();

        // ---- depthwise over E ----
        const int c = c0 + q * 4;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < Cexp) {
            const float4 (*wl)[8] = Wl[ALLC ? ch : ch & 1];
            float4 bi4;
            if constexpr (ALLC > 0) {
                bi4 = *reinterpret_cast<const float4*>(&E[(ch * 8 + q) * XD_LDE + XD_CC]);
            } else {
                bi4 = wl[KS * KS][q];
            }
            for (int st = sidx; st < NSTRIP; st += 32) {
                const int ro = st / SPR;
                const int so = (st - ro * SPR) * SW;
                float4 acc[SW];
#pragma unroll
                for (int o = 0; o < SW; ++o) acc[o] = bi4;
#pragma unroll 1   // a rolled filter-row loop keeps one row of weights live instead of KS*KS hoisted ones
                for (int r = 0; r < KS; ++r) {
                    float4 wr[KS];
#pragma unroll
                    for (int t = 0; t < KS; ++t) wr[t] = wl[r * KS + t][q];
                    const int rowbase = (ro * S + r) * IW;
                    float4 v[NIN];
#pragma unroll
                    for (int j = 0; j < NIN; ++j) {
                        int col = so * S + j;
                        if constexpr (TW % SW != 0) col = col < IW ? col : IW - 1;   // only outputs past the tile edge use these
                        v[j] = *reinterpret_cast<const float4*>(&E[(rowbase + col) * XD_LDE + q * 4]);
                    }
#pragma unroll
                    for (int j = 0; j < NIN; ++j)
#pragma unroll
                        for (int o = 0; o < SW; ++o) {
                            const int t = j - o * S;
                            if (t >= 0 && t < KS) {
                                acc[o].x = fmaf(v[j].x, wr[t].x, acc[o].x); acc[o].y = fmaf(v[j].y, wr[t].y, acc[o].y);
                                acc[o].z = fmaf(v[j].z, wr[t].z, acc[o].z); acc[o].w = fmaf(v[j].w, wr[t].w, acc[o].w);
                            }
                        }
                }
                const int ho = ho0 + ro;
#pragma unroll
                for (int o = 0; o < SW; ++o) {
                    const int wl = so + o, wo = wo0 + wl;
                    if (wl < TW && ho < Ho && wo < Wo) {
                        const xd_f32x2 ulo = xd_swish2(xd_f32x2{acc[o].x, acc[o].y}), uhi = xd_swish2(xd_f32x2{acc[o].z, acc[o].w});
                        const float4 u = make_float4(ulo.x, ulo.y, uhi.x, uhi.y);
                        const size_t yo = ybase + (unsigned)((ho * Wo + wo) * Cexp + c);
                        if (out16) {   // bf16 output tensor (round to nearest even): half the store traffic
                            const xd_f32x2 lo = {u.x, u.y}, hi = {u.z, u.w};
                            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y) + yo) =
                                make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, xd_bf16x2)),
                                           __builtin_bit_cast(unsigned, __builtin_convertvector(hi, xd_bf16x2)));
                        } else {
                            // streaming store: the tensor is far larger than the caches and is read next by another kernel
                            __builtin_nontemporal_store(xd_f32x4{u.x, u.y, u.z, u.w}, reinterpret_cast<xd_f32x4*>(y + yo));
                        }
                        sum.x += u.x; sum.y += u.y; sum.z += u.z; sum.w += u.w;
                    }
                }
            }
        }
        if (partial) {
            // per-channel sum over the tile in a fixed order: lanes with equal q inside a wave, then the four waves
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                sum.x += __shfl_xor(sum.x, off, 64); sum.y += __shfl_xor(sum.y, off, 64);
                sum.z += __shfl_xor(sum.z, off, 64); sum.w += __shfl_xor(sum.w, off, 64);
            }
            if (lane < 8) red[wave][lane] = sum;
        }
        __syncthreads();   // E is free for the next chunk; red is complete
        if (partial && tid < 8 && c < Cexp) {
            float4 t = red[0][tid];
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) { t.x += red[w2][tid].x; t.y += red[w2][tid].y; t.z += red[w2][tid].z; t.w += red[w2][tid].w; }
            *reinterpret_cast<float4*>(partial + ((size_t)b * gridDim.x + tile) * Cexp + c) = t;
        }
    }
}

struct XdCfg { int cin, ks, s, th, tw; };
// One tile shape per (input channels, filter, stride) of EfficientNet-B0's expanding blocks.
static const XdCfg kXd[] = {
    {16, 3, 2, 7, 8}, {24, 3, 1, 14, 14}, {24, 5, 2, 7, 7}, {40, 5, 1, 14, 14}, {40, 3, 2, 7, 7},
};

static const XdCfg* xd_find(int Cin, int KS, int stride) {
    for (const XdCfg& c : kXd)
        if (c.cin == Cin && c.ks == KS && c.s == stride) return &c;
    return nullptr;
}

// mbconv_late.hip: the whole-image kernel of the 14 x 14 / 7 x 7 stride-1 blocks (fp32 only; one sum per image and channel)
int expand_dwconv_whole_image(int H, int W, int Cin, int Cexp, int KS, int stride);
int expand_dwconv_img_f32(const float* x, const float* we, const float* es, const float* eb, const float* wd, const float* ds,
                          const float* db, float* y, float* pooled, int B, int H, int W, int Cin, int Cexp, int KS, int pad_t,
                          int pad_l, hipStream_t st);

// Number of per-image partial-sum chunks the fused kernel writes, or -1 when this layer shape has no fused kernel.
int expand_dwconv_chunks(int Ho, int Wo, int Cin, int Cexp, int KS, int stride) {
    if (expand_dwconv_whole_image(Ho, Wo, Cin, Cexp, KS, stride)) return 1;
    const XdCfg* c = xd_find(Cin, KS, stride);
    if (!c || Ho <= 0 || Wo <= 0 || Cexp <= 0 || Cexp % 4) return -1;
    return ((Ho + c->th - 1) / c->th) * ((Wo + c->tw - 1) / c->tw);
}

int expand_dwconv_f32(const float* x, const float* we, const float* es, const float* eb, const float* wd, const float* ds,
                      const float* db, float* y, float* partial, int B, int H, int W, int Cin, int Cexp, int KS, int stride,
                      int pad_t, int pad_l, int Ho, int Wo, int bf16, hipStream_t st) {
    if (!x || !we || !wd || !y || B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || pad_t < 0 || pad_l < 0) {
        set_error("expand_dwconv: bad arguments");
        return ICAP_ERR_ARG;
    }
    const int out16 = (bf16 >> 1) & 1;     // bf16: bit 0 = bf16 products, bit 1 = bf16 output tensor, bit 2 = fp32-accurate split products
    const int x3 = (bf16 >> 2) & 1;
    bf16 &= 1;
    if (x3 && (bf16 || out16)) { set_error("expand_dwconv: split products give an fp32 result"); return ICAP_ERR_ARG; }
    if (Ho == H && Wo == W && expand_dwconv_whole_image(H, W, Cin, Cexp, KS, stride)) {
        if (bf16 || out16) { set_error("expand_dwconv: the whole-image kernel of the 14x14 / 7x7 blocks is fp32 only"); return ICAP_ERR_ARG; }
        if (!aligned16(x) || !aligned16(we) || !aligned16(wd) || !aligned16(y) || (ds && !aligned16(ds)) || (db && !aligned16(db))) {
            set_error("expand_dwconv: pointers must be 16-byte aligned");
            return ICAP_ERR_ALIGN;
        }
        return expand_dwconv_img_f32(x, we, es, eb, wd, ds, db, y, partial, B, H, W, Cin, Cexp, KS, pad_t, pad_l, st);
    }
    const XdCfg* c = xd_find(Cin, KS, stride);
    if (!c || Cexp <= 0 || Cexp % 4) {
        set_error("expand_dwconv: no fused kernel for Cin=%d, %dx%d stride %d, Cexp=%d", Cin, KS, KS, stride, Cexp);
        return ICAP_ERR_ARG;
    }
    if ((long long)(Ho - 1) * stride - pad_t >= H || (long long)(Wo - 1) * stride - pad_l >= W) {
        set_error("expand_dwconv: output %dx%d does not fit input %dx%d", Ho, Wo, H, W);
        return ICAP_ERR_ARG;
    }
    if (!aligned16(x) || !aligned16(we) || !aligned16(wd) || !aligned16(y) || (partial && !aligned16(partial)) ||
        (ds && !aligned16(ds)) || (db && !aligned16(db))) {
        set_error("expand_dwconv: pointers must be 16-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    if (B > 65535) { set_error("expand_dwconv: B=%d exceeds 65535", B); return ICAP_ERR_LIMIT; }
    if ((long long)H * W * Cin >= (1ll << 31) || (long long)Ho * Wo * Cexp >= (1ll << 31) || (long long)Cexp * (Cin > KS * KS ? Cin : KS * KS) >= (1ll << 31)) {
        set_error("expand_dwconv: one image's input, output and the weights must each stay below 2^31 elements");
        return ICAP_ERR_LIMIT;
    }
    const int tiles_w = (Wo + c->tw - 1) / c->tw;
    const int ntiles = ((Ho + c->th - 1) / c->th) * tiles_w;
    const int nchunks = (Cexp + XD_CC - 1) / XD_CC;
    // enough workgroups to give every CU several: split the channel chunks over grid.z when tiles x images is small
    long long wg = (long long)ntiles * B;
    int zg = (int)((4096 + wg - 1) / wg);
    zg = zg < 1 ? 1 : (zg > nchunks ? nchunks : zg);
    const int cpg = (nchunks + zg - 1) / zg;
    zg = (nchunks + cpg - 1) / cpg;
    const dim3 grid(ntiles, B, zg), block(256);
#define ICAP_XD(CIN_, KS_, S_, TH_, TW_, SW_, ALLC_)                                                                                   \
    do {                                                                                                                          \
        if (x3 && (ALLC_) > 0 && zg == 1 && nchunks <= (ALLC_))                                                                    \
            hipLaunchKernelGGL((expand_dw_kernel<CIN_, KS_, S_, TH_, TW_, SW_, false, 3, ALLC_>), grid, block, 0, st, x, we, es,   \
                               eb, wd, ds, db, y, partial, H, W, Cexp, pad_t, pad_l, Ho, Wo, tiles_w, cpg, 0, 0, 0, 0, 0);         \
        else if (x3) hipLaunchKernelGGL((expand_dw_kernel<CIN_, KS_, S_, TH_, TW_, SW_, false, 3>), grid, block, 0, st, x, we, es, \
                                   eb, wd, ds, db, y, partial, H, W, Cexp, pad_t, pad_l, Ho, Wo, tiles_w, cpg, 0, 0, 0, 0, 0);     \
        else if (bf16) hipLaunchKernelGGL((expand_dw_kernel<CIN_, KS_, S_, TH_, TW_, SW_, false, 1>), grid, block, 0, st, x, we, es, \
                                     eb, wd, ds, db, y, partial, H, W, Cexp, pad_t, pad_l, Ho, Wo, tiles_w, cpg, 0, 0, 0, 0, out16); \
        else hipLaunchKernelGGL((expand_dw_kernel<CIN_, KS_, S_, TH_, TW_, SW_>), grid, block, 0, st, x, we, es, eb, wd, ds, db,  \
                                y, partial, H, W, Cexp, pad_t, pad_l, Ho, Wo, tiles_w, cpg, 0, 0, 0, 0, out16);                   \
    } while (0)
    if (Cin == 16) ICAP_XD(16, 3, 2, 7, 8, 2, 3);
    else if (Cin == 24 && KS == 3) ICAP_XD(24, 3, 1, 14, 14, 7, 0);
    else if (Cin == 24) ICAP_XD(24, 5, 2, 7, 7, 2, 0);
    else if (Cin == 40 && KS == 5) ICAP_XD(40, 5, 1, 14, 14, 7, 0);
    else ICAP_XD(40, 3, 2, 7, 7, 2, 0);
#undef ICAP_XD
    return launch_status("expand_dwconv");
}

// EfficientNet stem (3x3 stride 2 on the 4-channel-padded image, BN, swish) fused with block 0's depthwise 3x3 stride 1
// (+ BN + swish + squeeze-excite partial sums): x [B][img_h][img_w][4], w_stem [C][3][3][4], y [B][Ho][Wo][C].
constexpr int STEM_TH = 14, STEM_TW = 14;

int stem_dwconv_chunks(int Ho, int Wo, int C) {
    if (Ho <= 0 || Wo <= 0 || C <= 0 || C % 4) return -1;
    return ((Ho + STEM_TH - 1) / STEM_TH) * ((Wo + STEM_TW - 1) / STEM_TW);
}

int stem_dwconv_f32(const float* x, const float* ws, const float* ss, const float* sb, const float* wd, const float* ds,
                    const float* db, float* y, float* partial, int B, int img_h, int img_w, int Hs, int Ws, int C,
                    int stem_pad_t, int stem_pad_l, int pad_t, int pad_l, int Ho, int Wo, int bf16, hipStream_t st) {
    if (!x || !ws || !wd || !y || B <= 0 || img_h <= 0 || img_w <= 0 || Hs <= 0 || Ws <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 ||
        C % 4 || stem_pad_t < 0 || stem_pad_l < 0 || pad_t < 0 || pad_l < 0) {
        set_error("stem_dwconv: bad arguments");
        return ICAP_ERR_ARG;
    }
    if ((long long)(Hs - 1) * 2 - stem_pad_t >= img_h || (long long)(Ws - 1) * 2 - stem_pad_l >= img_w ||
        (long long)(Ho - 1) - pad_t >= Hs || (long long)(Wo - 1) - pad_l >= Ws) {
        set_error("stem_dwconv: output sizes do not fit the image");
        return ICAP_ERR_ARG;
    }
    if (!aligned16(x) || !aligned16(ws) || !aligned16(wd) || !aligned16(y) || (partial && !aligned16(partial)) ||
        (ds && !aligned16(ds)) || (db && !aligned16(db))) {
        set_error("stem_dwconv: pointers must be 16-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    if (B > 65535) { set_error("stem_dwconv: B=%d exceeds 65535", B); return ICAP_ERR_LIMIT; }
    if ((long long)Ho * Wo * C >= (1ll << 31)) { set_error("stem_dwconv: one image's output must stay below 2^31 elements"); return ICAP_ERR_LIMIT; }
    const int out16 = ((bf16 >> 1) & 1) | ((bf16 >> 3) & 1) << 1;     // bf16: bit 0 = bf16 products, bit 1 = bf16 output tensor, bit 3 = x is planar NCHW
    bf16 &= 1;
    const int tiles_w = (Wo + STEM_TW - 1) / STEM_TW;
    const int ntiles = ((Ho + STEM_TH - 1) / STEM_TH) * tiles_w;
    const int nchunks = (C + XD_CC - 1) / XD_CC;
    const dim3 grid(ntiles, B, 1), block(256);
    if (bf16) hipLaunchKernelGGL((expand_dw_kernel<40, 3, 1, STEM_TH, STEM_TW, 7, true, 1>), grid, block, 0, st, x, ws, ss, sb, wd, ds,
                                 db, y, partial, Hs, Ws, C, pad_t, pad_l, Ho, Wo, tiles_w, nchunks, img_h, img_w, stem_pad_t, stem_pad_l, out16);
    else hipLaunchKernelGGL((expand_dw_kernel<40, 3, 1, STEM_TH, STEM_TW, 7, true>), grid, block, 0, st, x, ws, ss, sb, wd, ds, db, y,
                            partial, Hs, Ws, C, pad_t, pad_l, Ho, Wo, tiles_w, nchunks, img_h, img_w, stem_pad_t, stem_pad_l, out16);
    return launch_status("stem_dwconv");
}

}  // namespace icap
