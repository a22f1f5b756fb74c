// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores, fp32 in,
// fp32 accumulate (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, so results
// stay within fp32 rounding of a CPU fp32 convolution).
//
//   y[m][n] = act( (sum_k A[m][k] * Wt[n][k]) * scale[n] + bias[n] + residual[m][n] )
//
// with m = (b, ho, wo) over NHWC activations, k = (r, q, c) over the filter
// window and A gathered on the fly (no im2col buffer).  A Linear layer is the
// 1x1 case with H = W = 1.  Weights are stored [Cout][KH][KW][Cin] so a filter
// row is K-contiguous.  An optional per-(image, input-channel) gate multiplies
// A while it is staged, which is how the squeeze-excite scale of an MBConv block
// is applied without a separate pass over the expanded tensor.
//
// Tiling: BM x BN outputs per 256-thread workgroup (4 waves), BK = 16.  Each wave
// owns a (BM/WM) x (BN/WN) sub-tile built from 32x32 MFMA blocks.  Operand tiles
// are staged global -> registers -> LDS (double buffered, one barrier per K step)
// with 80-byte LDS rows so that the ds_read_b128 fragment reads are conflict free.
#include "common.hpp"

namespace icap {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
    const float* x;
    const float* w;
    const float* scale;     // [Cout] or null (=1)
    const float* bias;      // [Cout] or null (=0)
    const float* residual;  // [M][ldy] or null
    const float* gate;      // [B][Cin] or null
    float* y;
    int B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, act;
    int M, K, ldy;
    int tiles_n;
};

constexpr int BK = 16;
constexpr int LDK = 20;  // padded LDS row (floats): 80 B keeps 16-B alignment, spreads banks

template <int BM, int BN, int WM, int WN, bool ONE_BY_ONE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int RM = TM / 32, RN = TN / 32;
    constexpr int RA = (BM * 4 + 255) / 256;  // float4 chunks of A per thread
    constexpr int RB = (BN * 4 + 255) / 256;
    static_assert(WM * WN == 4, "four waves per workgroup");
    static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile is built from 32x32 blocks");

    __shared__ __attribute__((aligned(16))) float As[2][BM][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN][LDK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;

    const int tile_n = blockIdx.x % p.tiles_n;
    const int tile_m = blockIdx.x / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread staging coordinates (fixed over the K loop) ----
    const int kc = tid & 3;  // which float4 of the 16-wide K slice
    const float* a_base[RA];
    int a_hi0[RA], a_wi0[RA], a_b[RA];
    bool a_ok[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int row = (tid >> 2) + 64 * i;
        const int m = m0 + row;
        a_ok[i] = (row < BM) && (m < p.M);
        const int mm = a_ok[i] ? m : 0;
        const int hw = p.Ho * p.Wo;
        const int b = mm / hw;
        const int rem = mm - b * hw;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        a_b[i] = b;
        a_hi0[i] = ho * p.stride - p.pad_t;
        a_wi0[i] = wo * p.stride - p.pad_l;
        a_base[i] = p.x + (size_t)b * p.H * p.W * p.Cin;
    }
    const float* b_base[RB];
    bool b_ok[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = (tid >> 2) + 64 * i;
        const int n = n0 + row;
        b_ok[i] = (row < BN) && (n < p.Cout);
        b_base[i] = p.w + (size_t)(b_ok[i] ? n : 0) * p.K;
    }

    float4 ra[RA], rb[RB];

    auto gload = [&](int t) {
        const int kk = t * BK + kc * 4;
        const bool kin = kk < p.K;
        int r = 0, q = 0, c = kk;
        if (!ONE_BY_ONE) {
            const int tap = kk / p.Cin;
            c = kk - tap * p.Cin;
            r = tap / p.KW;
            q = tap - r * p.KW;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int hi = a_hi0[i] + r, wi = a_wi0[i] + q;
            if (kin && a_ok[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
                v = *reinterpret_cast<const float4*>(a_base[i] + ((size_t)hi * p.W + wi) * p.Cin + c);
                if (p.gate) {
                    const float4 g = *reinterpret_cast<const float4*>(p.gate + (size_t)a_b[i] * p.Cin + c);
                    v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (kin && b_ok[i]) v = *reinterpret_cast<const float4*>(b_base[i] + kk);
            rb[i] = v;
        }
    };
    auto lds_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int row = (tid >> 2) + 64 * i;
            if (row < BM) *reinterpret_cast<float4*>(&As[buf][row][kc * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = (tid >> 2) + 64 * i;
            if (row < BN) *reinterpret_cast<float4*>(&Bs[buf][row][kc * 4]) = rb[i];
        }
    };

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    gload(0);
    lds_store(0);
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < nk; ++t) {
        const bool more = (t + 1) < nk;
        if (more) gload(t + 1);

        // Lane (lr, lh) owns k = 8*lh + s of this slice for MFMA step s: any
        // k -> (step, half) assignment is valid as long as A and B agree.
        float af[RM][8], bf[RN][8];
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const float* src = &As[cur][wm * TM + i * 32 + lr][8 * lh];
            const float4 v0 = *reinterpret_cast<const float4*>(src);
            const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
            af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
            af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
        }
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const float* src = &Bs[cur][wn * TN + j * 32 + lr][8 * lh];
            const float4 v0 = *reinterpret_cast<const float4*>(src);
            const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
            bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w;
            bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);

        if (more) lds_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds column n = lr, rows (e&3) + 8*(e>>2) + 4*lh ----
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int n = n0 + wn * TN + j * 32 + lr;
        if (n >= p.Cout) continue;
        const float sc = p.scale ? p.scale[n] : 1.f;
        const float bi = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m < p.M) {
                    float v = fmaf(acc[i][j][e], sc, bi);
                    const size_t o = (size_t)m * p.ldy + n;
                    if (p.residual) v += p.residual[o];
                    p.y[o] = act_apply(v, p.act);
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(ConvParams& p, hipStream_t st) {
    const int tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const long long blocks = (long long)tiles_m * p.tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        set_error("conv_igemm: grid of %lld workgroups out of range", blocks);
        return ICAP_ERR_LIMIT;
    }
    const bool one = (p.KH == 1 && p.KW == 1 && p.pad_t == 0 && p.pad_l == 0);
    if (one)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    return launch_status("conv_igemm");
}

int conv_igemm_f32(const float* x, const float* w, const float* scale, const float* bias,
                   const float* residual, const float* gate, float* y, int B, int H, int W, int Cin,
                   int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                   hipStream_t st) {
    if (!x || !w || !y) { set_error("conv_igemm: null tensor"); return ICAP_ERR_ARG; }
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 ||
        Ho <= 0 || Wo <= 0 || pad_t < 0 || pad_l < 0) {
        set_error("conv_igemm: non-positive dimension");
        return ICAP_ERR_ARG;
    }
    if (Cin % 4 != 0) { set_error("conv_igemm: Cin=%d must be a multiple of 4", Cin); return ICAP_ERR_ALIGN; }
    if (!aligned16(x) || !aligned16(w) || (gate && !aligned16(gate))) {
        set_error("conv_igemm: x, w and gate must be 16-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    // the last output position must start inside the padded input
    if ((long long)(Ho - 1) * stride - pad_t >= H || (long long)(Wo - 1) * stride - pad_l >= W) {
        set_error("conv_igemm: output %dx%d does not fit input %dx%d", Ho, Wo, H, W);
        return ICAP_ERR_ARG;
    }
    const long long M = (long long)B * Ho * Wo;
    const long long K = (long long)KH * KW * Cin;
    if (M > 0x7fffffffLL || K > 0x7fffffffLL || M * Cout > (1LL << 40)) {
        set_error("conv_igemm: problem too large");
        return ICAP_ERR_LIMIT;
    }
    ConvParams p{x, w, scale, bias, residual, gate, y, B, H, W, Cin, Cout, KH, KW,
                 stride, pad_t, pad_l, Ho, Wo, act, (int)M, (int)K, Cout, 0};
    // Tile choice: wide N tiles when there are channels to fill them, tall M tiles
    // for thin layers, and small tiles when the whole problem is only a few
    // hundred rows (decode-time linears) so that the grid still covers the chip.
    if (M <= 1024) {
        if (Cout <= 32) return launch_cfg<128, 32, 4, 1>(p, st);
        return launch_cfg<64, 64, 2, 2>(p, st);
    }
    if (Cout <= 32) return launch_cfg<256, 32, 4, 1>(p, st);
    if (Cout <= 64) return launch_cfg<256, 64, 4, 1>(p, st);
    return launch_cfg<128, 128, 2, 2>(p, st);
}

}  // namespace icap
