// Dense KH x KW convolutions of the ResNet-18 / VGG-16 trunks with bf16 activations end to end (bf16 NHWC in, bf16 NHWC out):
//   y[m][n] = act((sum_k x[m, k] w[n][k]) * scale[n] + bias[n] (+ residual[m][n])),  m = (b, oh, ow),  k = ((kh KW) + kw) Cin + c
// as an implicit GEMM on 256-row tiles with eight waves and NO operand byte through a register: every lane issues
// global_load_lds_dwordx4 (16 bytes global -> LDS; a wave-load fills 1 KB of consecutive LDS), the next 64-deep K tile is in
// flight while the current one multiplies, and the only synchronisation in the loop is one counted wait plus one raw
// s_barrier per K tile.  Cin % 64 == 0, so a 64-wide K tile lies inside one filter tap and a row's slice is one 128-byte line
// of the image; taps outside the image read a line of zeros instead (a per-lane source address costs nothing here).
// LDS image: [256 or BN rows][128 bytes], linear as the direct loads write it; the 16-byte slot a lane FETCHES is permuted
// (slot ^ ((row >> 1) & 7)) and the fragment reads apply the same permutation, which spreads the 16 rows of a ds_read_b128
// lane group over all 16 slots of the 256-byte bank row.
#include "common.hpp"

namespace icap {

typedef __bf16 cw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cw_f32x2 __attribute__((ext_vector_type(2)));
typedef float cw_f32x4 __attribute__((ext_vector_type(4)));
typedef float cw_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned cw_u32x4 __attribute__((ext_vector_type(4)));

struct ConvWideParams {
    const unsigned short* x;    // bf16 [B][H][W][Cin]
    const unsigned short* w;    // bf16 [Cout][KH][KW][Cin]
    const float* scale;
    const float* bias;
    const unsigned short* residual;   // bf16 [M][Cout] or null
    unsigned short* y;          // bf16 [M][Cout]
    int M, N, K, act, tiles_n, m_fast;
    int H, W, Cin, KW, stride, pad_t, pad_l, Ho, Wo;
};

constexpr int CW_BM = 256, CW_BK = 64, CW_ROWB = CW_BK * 2;      // tile rows, K tile, bytes per LDS row

// one line of zeros: the source of every tap outside the image
__device__ __attribute__((aligned(128))) unsigned char cw_zero_line[128];

__device__ __forceinline__ void cw_load16(const void* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ unsigned cw_pack(float a, float b) {
    const cw_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cw_bf16x2));
}

template <int BN>
constexpr int cw_smem_bytes() { return 2 * (CW_BM + BN) * CW_ROWB; }

// BN: tile width; WM x WN = 8 waves, each owning a (256 / WM) x (BN / WN) block of the tile
template <int BN, int WM, int WN>
__global__ __launch_bounds__(512, 1) void conv_bf16_wide_kernel(const ConvWideParams p) {
    static_assert(WM * WN == 8 && BN % 64 == 0, "eight waves; a pass of the eight waves' direct loads fills 64 rows");
    constexpr int BM = CW_BM, TM = BM / WM, TN = BN / WN, RM = TM / 32, RN = TN / 32;
    static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile is built from 32x32 blocks");
    constexpr int APT = BM / 64, BPT = BN / 64;                      // direct loads per lane and K tile
    constexpr int A_BYTES = BM * CW_ROWB, STAGE = (BM + BN) * CW_ROWB;
    constexpr int EP_LD = TN + 4;                                    // floats per row of a wave's result image
    static_assert(8 * 32 * EP_LD * 4 <= 2 * STAGE, "the eight waves' result images fit the staging buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char cw_smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lr = lane & 31, lh = lane >> 5;

    // XCD x walks a contiguous range of the tile list; neighbours in time share the larger operand's panel
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tq = nwg >> 3, tr = nwg & 7;
    const int tlin = xcd * tq + (xcd < tr ? xcd : tr) + slot;
    int tile_m, tile_n;
    if (p.m_fast) {
        const int tiles_m = nwg / p.tiles_n;
        tile_n = tlin / tiles_m;
        tile_m = tlin - tile_n * tiles_m;
    } else {
        tile_m = tlin / p.tiles_n;
        tile_n = tlin - tile_m * p.tiles_n;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- direct loads: pass i of the eight waves fills rows 64 i .. 64 i + 63; this lane's row in a pass and the slot it fetches
    const int srow = wave * 8 + (lane >> 3);
    const int lslot = (lane & 7) ^ ((wave * 4 + (lane >> 4)) & 7);   // = physical slot ^ ((row >> 1) & 7); 64 i does not change it
    const unsigned short* a_base[APT];       // pixel (b, oh stride - pad_t, ow stride - pad_l), channel 8 lslot (may lie outside the image)
    int a_ih[APT], a_iw[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int m = m0 + i * 64 + srow;
        const int mm = min(m, p.M - 1);
        const int b = mm / (p.Ho * p.Wo), rem = mm - b * (p.Ho * p.Wo);
        const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
        const int ih0 = oh * p.stride - p.pad_t;
        a_iw[i] = ow * p.stride - p.pad_l;
        a_base[i] = p.x + (((long long)b * p.H + ih0) * p.W + a_iw[i]) * p.Cin + 8 * lslot;
        a_ih[i] = m < p.M ? ih0 : -(1 << 20);                          // rows past the matrix read zeros
    }
    const unsigned short* b_src[BPT];
#pragma unroll
    for (int i = 0; i < BPT; ++i) b_src[i] = p.w + (size_t)min(n0 + i * 64 + srow, p.N - 1) * p.K + 8 * lslot;
    const unsigned char* zsrc = cw_zero_line + 16 * (lane & 7);

    int kh = 0, kw = 0, c0 = 0;              // filter tap and first channel of the K tile issue() fetches next
    auto issue = [&](int t, int buf) __attribute__((always_inline)) {
        unsigned char* base = cw_smem + buf * STAGE + wave * 1024;
        const long long tap_off = ((long long)kh * p.W + kw) * p.Cin + c0;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int ih = a_ih[i] + kh, iw = a_iw[i] + kw;
            const bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            // a select on the address, never a branch around the load (branches would serialise the four loads)
            const unsigned long long real = reinterpret_cast<unsigned long long>(a_base[i] + tap_off), zero = reinterpret_cast<unsigned long long>(zsrc);
            const unsigned long long src = zero + ((real - zero) & (0ull - (unsigned long long)ok));
            cw_load16(reinterpret_cast<const void*>(src), base + i * 8192);
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) cw_load16(b_src[i] + (size_t)t * CW_BK, base + A_BYTES + i * 8192);
        c0 += CW_BK;
        if (c0 == p.Cin) { c0 = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
    };

    cw_f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addresses: row r, logical slot q at r * 128 + ((q ^ ((r >> 1) & 7)) * 16); rows here are lr + multiples of 32
    const int sw = (lr >> 1) & 7;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* sa = cw_smem + buf * STAGE + (wm * TM + lr) * CW_ROWB;
        const unsigned char* sb = cw_smem + buf * STAGE + A_BYTES + (wn * TN + lr) * CW_ROWB;
#pragma unroll
        for (int s = 0; s < CW_BK / 16; ++s) {
            const int off = ((2 * s + lh) ^ sw) * 16;
            cw_bf16x8 af[RM], bf[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i) af[i] = *reinterpret_cast<const cw_bf16x8*>(sa + i * 32 * CW_ROWB + off);
#pragma unroll
            for (int j = 0; j < RN; ++j) bf[j] = *reinterpret_cast<const cw_bf16x8*>(sb + j * 32 * CW_ROWB + off);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = p.K / CW_BK;
    issue(0, 0);
    for (int t = 0; t < nk; ++t) {
        // K tile t has landed; the barrier makes every wave's part of it visible and says that everybody is done reading
        // the buffer the next tile's loads overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 1 < nk) issue(t + 1, (t + 1) & 1);
        compute(t & 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // every wave is done with the staging buffers: the result images overwrite them
    asm volatile("" ::: "memory");

    // ---- epilogue, every wave on its own, one 32-row block at a time: accumulators -> wave-private LDS image (row-major fp32)
    // -> 8 columns per lane: BN, activation, residual, 16-byte bf16 stores along the output row
    float* ew = reinterpret_cast<float*>(cw_smem) + wave * 32 * EP_LD;
    constexpr int PPR = TN / 8;                        // 16-byte output pieces per row
    constexpr int NPC = 32 * PPR / 64;                 // pieces per lane and row block
#pragma unroll
    for (int i = 0; i < RM; ++i) {
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) ew[((e & 3) + 8 * (e >> 2) + 4 * lh) * EP_LD + j * 32 + lr] = acc[i][j][e];
#pragma unroll
        for (int q = 0; q < NPC; ++q) {
            const int pc = lane + 64 * q, row = pc / PPR, c8 = pc - row * PPR;
            const int m = m0 + wm * TM + i * 32 + row, n = n0 + wn * TN + 8 * c8;
            const cw_f32x4 v0 = *reinterpret_cast<const cw_f32x4*>(ew + row * EP_LD + 8 * c8);
            const cw_f32x4 v1 = *reinterpret_cast<const cw_f32x4*>(ew + row * EP_LD + 8 * c8 + 4);
            if (m >= p.M || n >= p.N) continue;
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.scale) {
                const cw_f32x4 s0 = *reinterpret_cast<const cw_f32x4*>(p.scale + n), s1 = *reinterpret_cast<const cw_f32x4*>(p.scale + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] *= s0[e]; v[4 + e] *= s1[e]; }
            }
            if (p.bias) {
                const cw_f32x4 b0 = *reinterpret_cast<const cw_f32x4*>(p.bias + n), b1 = *reinterpret_cast<const cw_f32x4*>(p.bias + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
            }
            const size_t o = (size_t)m * p.N + n;
            if (p.residual) {
                const cw_u32x4 r = *reinterpret_cast<const cw_u32x4*>(p.residual + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += __uint_as_float(r[e] << 16);
                    v[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u);
                }
            }
            if (p.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.act != ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e], p.act);
            }
            cw_u32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) out[e] = cw_pack(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<cw_u32x4*>(p.y + o) = out;
        }
    }
}

template <int BN, int WM, int WN>
static int cw_launch(ConvWideParams& p, hipStream_t st) {
    static hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_wide_kernel<BN, WM, WN>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, cw_smem_bytes<BN>());
    if (attr_rc != hipSuccess) { set_error("conv2d_bf16 (wide): cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr_rc)); return ICAP_ERR_HIP_BASE + (int)attr_rc; }
    p.tiles_n = (p.N + BN - 1) / BN;
    const long long wgs = (long long)((p.M + CW_BM - 1) / CW_BM) * p.tiles_n;
    if (wgs > 0x7fffffffLL) { set_error("conv2d_bf16 (wide): problem too large"); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL((conv_bf16_wide_kernel<BN, WM, WN>), dim3((unsigned)wgs), dim3(512), cw_smem_bytes<BN>(), st, p);
    return launch_status("conv2d_bf16 (wide)");
}

// 1 when the wide form runs this layer: bf16 image in and out, whole 64-channel K tiles, 16-byte output pieces
int conv2d_bf16_wide_supported(int Cin, int Cout) { return Cin > 0 && Cin % 64 == 0 && Cout > 0 && Cout % 8 == 0; }

// the 256-row form of conv2d_bf16_f32 (its cfg 5): x, y and residual (optional) bf16
int conv2d_bf16_wide(const unsigned short* x, const unsigned short* w, const float* scale, const float* bias, const unsigned short* residual,
                     unsigned short* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho,
                     int Wo, int act, hipStream_t st) {
    if (!conv2d_bf16_wide_supported(Cin, Cout)) {
        set_error("conv2d_bf16 (wide): Cin=%d must be a multiple of 64 and Cout=%d of 8", Cin, Cout);
        return ICAP_ERR_ARG;
    }
    const long long M = (long long)B * Ho * Wo, K = (long long)KH * KW * Cin;
    if (M > 0x7fffffffLL || K > 0x7fffffffLL) { set_error("conv2d_bf16 (wide): problem too large"); return ICAP_ERR_LIMIT; }
    ConvWideParams p{x, w, scale, bias, residual, y, (int)M, Cout, (int)K, act, 0, 0, H, W, Cin, KW, stride, pad_t, pad_l, Ho, Wo};
    p.m_fast = 0;          // consecutive tiles of an XCD share their image rows and walk over the column tiles: the CUs of an XCD
                           // then multiply neighbouring image rows by the same (L2-resident) weights
    if (Cout > 128) return cw_launch<256, 2, 4>(p, st);
    if (Cout > 64) return cw_launch<128, 4, 2>(p, st);
    return cw_launch<64, 4, 2>(p, st);
}

}  // namespace icap
