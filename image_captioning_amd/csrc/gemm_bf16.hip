// Linear layers / 1x1 convolutions on the bf16 matrix cores with fp32 activations in memory:
//   y[M][N] = act((x[M][K] (* gate[m / hw][k]) @ w[N][K]^T) * scale[n] + bias[n] (+ residual[m][n]))
// x, y, scale, bias, residual, gate are fp32; w is bf16 (converted once when the weights are prepared).  The A operand
// is rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32) while it is staged into LDS, products are accumulated
// in fp32 by v_mfma_f32_32x32x16_bf16 - the "bf16 compute, fp32 storage" mode of the captioning path.
//
// A workgroup (4 waves, 2 x 2) owns a BM x BN output tile; operand tiles of 32 K elements are staged global ->
// registers -> LDS (double buffered, one barrier per step), LDS rows are padded to 80 bytes so that the 16-byte
// fragment reads (lane (r, h) reads k = 8h .. 8h+7 of row r) are conflict free.  The epilogue transposes the
// accumulators through LDS and writes 16 bytes per lane along the output row.
#include <type_traits>
#include "common.hpp"

namespace icap {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2b __attribute__((ext_vector_type(2)));
typedef float f32x16b __attribute__((ext_vector_type(16)));

struct GemmBf16Params {
    const float* x;
    const unsigned short* w;
    const float* scale;
    const float* bias;
    const float* residual;
    const float* gate;      // [M / hw][K] or null
    float* y;               // may be null when only the arg-max partials are wanted
    int M, N, K, hw, act, tiles_n;
    float* amax_val;        // optional [M][tiles_n]: per-tile row maxima of the outputs (arg-max partials)
    int* amax_idx;          // their column indices (first index on ties)
    // CONV kernels only: x is an NHWC image [B][H][W][Cin], row m = (b, oh, ow), k = ((kh * KW) + kw) * Cin + c
    int H, W, Cin, KW, stride, pad_t, pad_l, Ho, Wo, KHW;
    int y16, res16;         // y is stored / residual is read as bf16 [M][N] (activations kept in bf16 between layers)
    int m_fast;             // tile order inside an XCD's range: 1 = consecutive tiles share a W panel, 0 = an A panel
    float* stat_part;       // optional [P][2][N]: column sum and squared deviations of every BM-row tile of the result
    int ksl;                // linear form only: K slice per blockIdx.z (0 = the whole contraction); slice z writes its partial
                            // products to y + z M N (no scale / bias / residual / activation: the reduction applies them)
};

// Workgroups go to the 8 XCDs round-robin by blockIdx and every XCD has its own L2.  Mapped naively (tile = blockIdx), the
// tiles that share an operand panel land on 8 different L2s and each of them fetches the panel from memory (measured on the
// 1024 x 512 x 6756 vocabulary projection: 184 MB fetched per launch for 23 MB of operands).  Instead XCD x works through a
// contiguous range of the tile list, ordered so that neighbours in time share the larger operand's panel.
__device__ __forceinline__ void gb16_tile(const GemmBf16Params& p, int& tile_m, int& tile_n) {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int t = xcd * q + (xcd < r ? xcd : r) + slot;     // XCD x owns q (+1 for x < r) consecutive tiles
    if (p.m_fast) {
        const int tiles_m = nwg / p.tiles_n;
        tile_n = t / tiles_m;
        tile_m = t - tile_n * tiles_m;
    } else {
        tile_m = t / p.tiles_n;
        tile_n = t - tile_m * p.tiles_n;
    }
}

// host: which operand is the larger one (bytes per element of x and w; taps = filter taps sharing one input pixel)
static inline void gb16_set_order(GemmBf16Params& p, int x_bytes, int w_bytes, int taps = 1) {
    p.m_fast = (long long)p.N * w_bytes * taps > (long long)p.M * x_bytes;
}

constexpr int GB16_BK = 32;             // K elements per step (two MFMA K blocks)
constexpr int GB16_LDK = GB16_BK + 8;   // padded LDS row, bf16 elements (80 bytes)

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2b v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// A16: x is a bf16 tensor (the output of a fused kernel run with a bf16 result): its tile goes to LDS as loaded, or
// through an fp32 multiply when the squeeze-excite gate applies.
// CONV: implicit GEMM of a KH x KW convolution: the A rows are gathered from the image on the fly, taps outside the image
// contribute zeros.  CONV = 1: Cin a multiple of 32, so a 32-wide K slice lies inside one filter tap.  CONV = 2: Cin = 4
// (the channel-padded RGB image): a slice covers 8 taps, one per staging lane; the weight rows are zero-padded to a
// multiple of 32 and taps beyond KH * KW read as zeros.
//
// SPLIT = 3: fp32-accurate products on the bf16 matrix cores.  Every fp32 operand is the exact sum of three bf16 values
// (hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid): 3 x 8 mantissa bits cover fp32's 24), so
//   a * b = hi.hi + (hi.mid + mid.hi) + (mid.mid + hi.lo + lo.hi) + [mid.lo + lo.mid + lo.lo, each <= 2^-24 |a b|, dropped]
// and the six kept bf16 products are exact in the MFMA's fp32 accumulation: the result differs from an fp32 dot product
// by a few units of fp32 rounding, at 6/16 of the fp32 MFMA's cost per multiply-add (v_mfma_f32_32x32x2_f32 retires
// 64 flop per cycle and SIMD, v_mfma_f32_32x32x16_bf16 1024).  x is split while it is staged (three LDS planes), w points
// to three pre-split planes [3][N][K].
// With A16 the activations arrive pre-split too ([3][M][K] planes from split3_f32): a GEMM with many column tiles (the
// decoder's vocabulary projection: 53) would otherwise repeat the split arithmetic of x in every one of them.
// Its three operand planes are single-buffered in LDS (60 KB for a 128 x 128 x 32 tile) so that TWO workgroups share a CU:
// one workgroup's staging phase (global-load wait, the split arithmetic, LDS stores) runs under the other's MFMA phase.
// STATS: the epilogue also leaves the column statistics of the result for a train-mode BatchNorm (its own instantiation, so
// that the registers its moments take are paid by those launches only).
template <int BM, int BN, bool GATE, bool AMAX = false, bool A16 = false, int CONV = 0, int SPLIT = 1, bool STATS = false>
__global__ __launch_bounds__(256, SPLIT == 3 ? 2 : 1) void gemm_bf16_kernel(const GemmBf16Params p) {
    static_assert(!(CONV && (GATE || AMAX)) && !(CONV == 2 && A16), "the convolution form has no gate / arg-max; its 4-channel form reads fp32");
    static_assert(SPLIT == 1 || (SPLIT == 3 && CONV == 0 && !(A16 && GATE)), "split products: linear form; pre-split activations take no gate");
    constexpr int BK = GB16_BK, LDK = GB16_LDK;
    constexpr int TM = BM / 2, TN = BN / 2, RM = TM / 32, RN = TN / 32;
    constexpr int AP = BM / 32;                    // staging passes over A: thread t -> row t/8 + 32 i, floats 4 (t%8) ..
    constexpr int BP = (BN + 63) / 64;             // staging passes over W: thread t -> row t/4 + 64 i, bf16 8 (t%4) ..
    constexpr int EP_LD = BN + 4;
    constexpr int EP_ROWS = BM < 64 ? BM : 64;
    constexpr int NBUF = SPLIT == 3 ? 1 : 2;        // LDS buffers per operand plane
    constexpr int STAGE_BYTES = SPLIT * NBUF * (BM + BN) * LDK * 2;
    constexpr int EP_BYTES = EP_ROWS * EP_LD * 4;
    constexpr int SMEM_BYTES = STAGE_BYTES > EP_BYTES ? STAGE_BYTES : EP_BYTES;
    static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile is built from 32x32 blocks");

    // the split form's three planes exceed the 64 KB of static LDS: it runs on dynamic LDS (gb16_smem_bytes)
    __shared__ __attribute__((aligned(16))) unsigned char smem_static[SPLIT == 1 ? SMEM_BYTES : 16];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    unsigned char* smem = SPLIT == 1 ? smem_static : smem_dyn;
    unsigned short* As = reinterpret_cast<unsigned short*>(smem);            // [SPLIT][NBUF][BM][LDK]
    unsigned short* Bs = As + SPLIT * NBUF * BM * LDK;                       // [SPLIT][NBUF][BN][LDK]
    constexpr int APL = NBUF * BM * LDK, BPL = NBUF * BN * LDK;              // elements per plane
    float* ep = reinterpret_cast<float*>(smem);                              // [EP_ROWS][EP_LD], used after the K loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    int tile_m, tile_n;
    gb16_tile(p, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int koff = p.ksl ? (int)blockIdx.z * p.ksl : 0;                  // this workgroup's K slice [koff, koff + klen)
    const int klen = p.ksl ? min(p.ksl, p.K - koff) : p.K;
    float* const y_out = p.y ? p.y + (p.ksl ? (size_t)blockIdx.z * p.M * p.N : 0) : nullptr;

    // staging coordinates
    const int a_k4 = (tid & 7) * 4, a_row = tid >> 3;
    // W staging: four lanes cover a row's 64 bytes.  An LDS store instruction is serviced in groups of 8 lanes and rows are 80
    // bytes apart, so the two rows of a group must be 4 rows apart (4 x 20 dwords = 16 mod 32 banks) not to meet in a bank.
    const int b_k8 = (tid & 3) * 8, b_q = tid >> 2, b_row = (b_q & 1) * 4 + ((b_q >> 1) & 3) + (b_q >> 3) * 8;
    const float* a_src[AP];
    const float* g_src[GATE ? AP : 1];
    bool a_ok[AP];
    int a_ih[CONV ? AP : 1], a_iw[CONV ? AP : 1];      // CONV: input coordinates of filter tap (0, 0) for this row
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int m = m0 + a_row + 32 * i;
        a_ok[i] = m < p.M;
        if constexpr (CONV) {
            const int mm = a_ok[i] ? m : 0;
            const int b = mm / (p.Ho * p.Wo), r = mm - b * (p.Ho * p.Wo);
            const int oh = r / p.Wo, ow = r - oh * p.Wo;
            a_ih[i] = oh * p.stride - p.pad_t;
            a_iw[i] = ow * p.stride - p.pad_l;
            a_src[i] = p.x + (size_t)b * p.H * p.W * p.Cin;
        } else {
            a_src[i] = p.x + (size_t)(a_ok[i] ? m : 0) * p.K + koff;
        }
        if (GATE) g_src[i] = p.gate + (size_t)((a_ok[i] ? m : 0) / p.hw) * p.K + koff;
    }
    int c_kh = 0, c_kw = 0, c_c0 = 0;                   // CONV: filter tap and first channel of the slice gload fetches next
    // bf16 A: staged like W (thread t -> row t/4 + 64 i, eight bf16 at 8 (t%4))
    constexpr int AP16 = (BM + 63) / 64;
    const unsigned short* a16_src[A16 ? AP16 : 1];
    const float* g16_src[(A16 && GATE) ? AP16 : 1];
    bool a16_ok[A16 ? AP16 : 1];
    int a16_ih[(A16 && CONV) ? AP16 : 1], a16_iw[(A16 && CONV) ? AP16 : 1];
    if constexpr (A16) {
#pragma unroll
        for (int i = 0; i < AP16; ++i) {
            const int m = m0 + b_row + 64 * i;
            a16_ok[i] = (b_row + 64 * i < BM) && m < p.M;
            if constexpr (CONV) {
                const int mm = a16_ok[i] ? m : 0;
                const int b = mm / (p.Ho * p.Wo), r = mm - b * (p.Ho * p.Wo);
                const int oh = r / p.Wo, ow = r - oh * p.Wo;
                a16_ih[i] = oh * p.stride - p.pad_t;
                a16_iw[i] = ow * p.stride - p.pad_l;
                a16_src[i] = reinterpret_cast<const unsigned short*>(p.x) + (size_t)b * p.H * p.W * p.Cin;
            } else {
                a16_src[i] = reinterpret_cast<const unsigned short*>(p.x) + (size_t)(a16_ok[i] ? m : 0) * p.K + koff;
            }
            if constexpr (GATE) g16_src[i] = p.gate + (size_t)((a16_ok[i] ? m : 0) / p.hw) * p.K + koff;
        }
    }
    const unsigned short* b_src[BP];
    bool b_ok[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) {
        const int n = n0 + b_row + 64 * i;
        b_ok[i] = (b_row + 64 * i < BN) && n < p.N;
        b_src[i] = p.w + (size_t)(b_ok[i] ? n : 0) * p.K + koff;
    }

    constexpr int NSET = 1;                          // staging register sets (K slices in flight)
    float4 ra[NSET][AP], rg[NSET][GATE ? AP : 1];
    uint4 rb[NSET][SPLIT][BP];
    uint4 ra16[A16 ? SPLIT : 1][A16 ? AP16 : 1];
    float4 rg16[(A16 && GATE) ? AP16 : 1][2];
    unsigned okmask_s[NSET];
    // every load is unconditional (clamped address); out-of-range elements are zeroed when the slice goes to LDS
    // FAST (a whole slice inside the contraction, linear form): nothing is masked.  Rows beyond M or N read row 0 (the
    // clamped pointers above) instead of zeros; their products land in output rows / columns that the epilogue never stores.
    auto gload = [&](int t, auto set_tag, auto fast_tag) {
        constexpr int S = decltype(set_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;
        unsigned okmask = 0;
        const int ka = t * BK + a_k4, kb = t * BK + b_k8;
        if constexpr (A16) {
#pragma unroll
            for (int i = 0; i < AP16; ++i) {
                bool ok = FAST || (a16_ok[i] && kb < klen);
                const unsigned short* src = a16_src[i] + kb;
                if constexpr (CONV == 1) {
                    const int ih = a16_ih[i] + c_kh, iw = a16_iw[i] + c_kw;
                    ok = a16_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    src = a16_src[i] + ((size_t)ih * p.W + iw) * p.Cin + c_c0 + b_k8;
                }
#pragma unroll
                for (int sp = 0; sp < SPLIT; ++sp)
                    ra16[sp][i] = *reinterpret_cast<const uint4*>(ok ? src + (size_t)sp * p.M * p.K : reinterpret_cast<const unsigned short*>(p.x));
                if constexpr (GATE) {
                    rg16[i][0] = *reinterpret_cast<const float4*>(ok ? g16_src[i] + kb : p.gate);
                    rg16[i][1] = *reinterpret_cast<const float4*>(ok ? g16_src[i] + kb + 4 : p.gate);
                }
                okmask |= ok ? (1u << i) : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < (A16 ? 0 : AP); ++i) {
            if constexpr (CONV == 1) {
                const int ih = a_ih[i] + c_kh, iw = a_iw[i] + c_kw;
                const bool ok = a_ok[i] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                ra[S][i] = *reinterpret_cast<const float4*>(ok ? a_src[i] + ((size_t)ih * p.W + iw) * p.Cin + c_c0 + a_k4 : p.x);
                okmask |= ok ? (1u << i) : 0u;
            } else if constexpr (CONV == 2) {
                const int tap = (ka >> 2), kh = tap / p.KW, kw = tap - kh * p.KW;
                const int ih = a_ih[i] + kh, iw = a_iw[i] + kw;
                const bool ok = a_ok[i] && tap < p.KHW && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                ra[S][i] = *reinterpret_cast<const float4*>(ok ? a_src[i] + ((size_t)ih * p.W + iw) * 4 : p.x);
                okmask |= ok ? (1u << i) : 0u;
            } else {
                const bool ok = FAST || (a_ok[i] && ka < klen);
                ra[S][i] = *reinterpret_cast<const float4*>(ok ? a_src[i] + ka : p.x);
                if (GATE) rg[S][i] = *reinterpret_cast<const float4*>(ok ? g_src[i] + ka : p.gate);
                okmask |= ok ? (1u << i) : 0u;
            }
        }
        if constexpr (CONV == 1) {  // slices are fetched in order: step the tap
            c_c0 += BK;
            if (c_c0 == p.Cin) {
                c_c0 = 0;
                if (++c_kw == p.KW) { c_kw = 0; ++c_kh; }
            }
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const bool ok = FAST || (b_ok[i] && kb < klen);
#pragma unroll
            for (int sp = 0; sp < SPLIT; ++sp)
                rb[S][sp][i] = *reinterpret_cast<const uint4*>(ok ? b_src[i] + (size_t)sp * p.N * p.K + kb : p.w);
            okmask |= ok ? (1u << (16 + i)) : 0u;
        }
        okmask_s[S] = okmask;
    };
    auto lds_store = [&](int buf, auto set_tag, auto fast_tag) {
        constexpr int S = decltype(set_tag)::value;
        const unsigned okmask = decltype(fast_tag)::value ? 0xffffffffu : okmask_s[S];
        if constexpr (A16) {
#pragma unroll
            for (int i = 0; i < AP16; ++i) {
                if (BM % 64 == 0 || b_row + 64 * i < BM) {
#pragma unroll
                    for (int sp = 1; sp < SPLIT; ++sp)
                        *reinterpret_cast<uint4*>(&As[sp * APL + (buf * BM + b_row + 64 * i) * LDK + b_k8]) =
                            ((okmask >> i) & 1u) ? ra16[sp][i] : make_uint4(0u, 0u, 0u, 0u);
                    uint4 o = ra16[0][i];
                    if constexpr (GATE) {   // bf16 -> fp32 is a 16-bit shift; multiply by the gate, round back
                        const unsigned w[4] = {o.x, o.y, o.z, o.w};
                        const float g[8] = {rg16[i][0].x, rg16[i][0].y, rg16[i][0].z, rg16[i][0].w,
                                            rg16[i][1].x, rg16[i][1].y, rg16[i][1].z, rg16[i][1].w};
                        unsigned r[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            r[j] = pack_bf16(__uint_as_float(w[j] << 16) * g[2 * j], __uint_as_float(w[j] & 0xffff0000u) * g[2 * j + 1]);
                        o = make_uint4(r[0], r[1], r[2], r[3]);
                    }
                    if (!((okmask >> i) & 1u)) o = make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(&As[(buf * BM + b_row + 64 * i) * LDK + b_k8]) = o;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < (A16 ? 0 : AP); ++i) {
            float4 v = ra[S][i];
            if (GATE) { v.x *= rg[S][i].x; v.y *= rg[S][i].y; v.z *= rg[S][i].z; v.w *= rg[S][i].w; }
            if (!((okmask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            uint2 o = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
            *reinterpret_cast<uint2*>(&As[(buf * BM + a_row + 32 * i) * LDK + a_k4]) = o;
            if constexpr (SPLIT == 3) {
                // residual after the bf16 head (exact in fp32), its bf16 head, and the residual after that
                float4 r = make_float4(v.x - __uint_as_float(o.x << 16), v.y - __uint_as_float(o.x & 0xffff0000u),
                                       v.z - __uint_as_float(o.y << 16), v.w - __uint_as_float(o.y & 0xffff0000u));
                const uint2 m = make_uint2(pack_bf16(r.x, r.y), pack_bf16(r.z, r.w));
                *reinterpret_cast<uint2*>(&As[APL + (buf * BM + a_row + 32 * i) * LDK + a_k4]) = m;
                r = make_float4(r.x - __uint_as_float(m.x << 16), r.y - __uint_as_float(m.x & 0xffff0000u),
                                r.z - __uint_as_float(m.y << 16), r.w - __uint_as_float(m.y & 0xffff0000u));
                *reinterpret_cast<uint2*>(&As[2 * APL + (buf * BM + a_row + 32 * i) * LDK + a_k4]) = make_uint2(pack_bf16(r.x, r.y), pack_bf16(r.z, r.w));
            }
        }
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            if (BN % 64 == 0 || b_row + 64 * i < BN) {
#pragma unroll
                for (int sp = 0; sp < SPLIT; ++sp) {
                    const uint4 o = ((okmask >> (16 + i)) & 1u) ? rb[S][sp][i] : make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(&Bs[sp * BPL + (buf * BN + b_row + 64 * i) * LDK + b_k8]) = o;
                }
            }
        }
    };

    f32x16b acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int buf) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const int koff = s * 16 + 8 * lh;
            bf16x8 af[SPLIT][RM], bf[SPLIT][RN];
#pragma unroll
            for (int sp = 0; sp < SPLIT; ++sp) {
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    af[sp][i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&As[sp * APL + (buf * BM + wm * TM + i * 32 + lr) * LDK + koff]));
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    bf[sp][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&Bs[sp * BPL + (buf * BN + wn * TN + j * 32 + lr) * LDK + koff]));
            }
            if constexpr (SPLIT == 3) {
                // smallest terms first: (a plane, b plane) pairs with plane sum <= 2.  Product-major order: consecutive MFMAs go
                // to different accumulators, so none waits for the previous one's result
                constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < RM; ++i)
#pragma unroll
                        for (int j = 0; j < RN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa[q]][i], bf[pb[q]][j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
            }
        }
    };

    const int nk = (klen + BK - 1) / BK;
    using T0 = std::integral_constant<int, 0>;
    using Slow = std::false_type;
    if constexpr (NBUF == 1) {
        // one LDS buffer: [wait for slice t's registers, split, store] [barrier] [fetch slice t+1] [MFMAs] [barrier]; the
        // CU's other workgroup fills the gaps
        auto k_loop = [&](auto fast_tag) {
            gload(0, T0{}, fast_tag);
            for (int t = 0; t < nk; ++t) {
                lds_store(0, T0{}, fast_tag);
                __syncthreads();
                if (t + 1 < nk) gload(t + 1, T0{}, fast_tag);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(2);      // the matrix-core phase outranks the other workgroup's staging arithmetic
                compute(0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
            }
        };
        // the column statistics count every row of the tile, so that form keeps the zeroed rows
        if (!STATS && CONV == 0 && klen % BK == 0) k_loop(std::true_type{});
        else k_loop(Slow{});
    } else {
        gload(0, T0{}, Slow{});
        lds_store(0, T0{}, Slow{});
        __syncthreads();
        int cur = 0;
        for (int t = 0; t < nk; ++t) {
            const bool more = t + 1 < nk;
            if (more) gload(t + 1, T0{}, Slow{});
            compute(cur);
            if (more) lds_store(cur ^ 1, T0{}, Slow{});
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue: accumulators -> LDS (row-major) -> 16-byte stores along the output row ----
    constexpr int BN4 = BN / 4;
    constexpr int EP_RPI = 256 / BN4;              // rows covered per pass of the workgroup
    constexpr int NPASS = BM / EP_ROWS;
    const int en4 = tid % BN4, erow = tid / BN4;
    const int n = n0 + en4 * 4;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), bi4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < p.N) {
        if (p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + n);
        if (p.bias) bi4 = *reinterpret_cast<const float4*>(p.bias + n);
    }
    static_assert(EP_ROWS * EP_LD >= 2 * 4 * EP_RPI * BN4, "the epilogue tile image also holds the statistics scratch");
    // stat_part: moments of this thread's rows (four columns) about the tile's first row, carried over the passes
    float4 st_shift = make_float4(0.f, 0.f, 0.f, 0.f), st_s1 = st_shift, st_s2 = st_shift;
    float my_shift = 0.f;
    dispatch_act(p.act, [&](auto tag) {
        constexpr int ACT = decltype(tag)::value;
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int r0 = pass * EP_ROWS;
            if (wm * TM >= r0 && wm * TM < r0 + EP_ROWS) {
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int row = wm * TM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh - r0;
                            ep[row * EP_LD + wn * TN + j * 32 + lr] = acc[i][j][e];
                        }
            }
            __syncthreads();
            if (STATS && pass == 0) {
                st_shift = *reinterpret_cast<const float4*>(&ep[en4 * 4]);        // the tile's first row: a sample of every column
                if (tid < BN) my_shift = ep[tid];
            }
            for (int row = erow; row < EP_ROWS; row += EP_RPI) {
                const int m = m0 + r0 + row;
                if constexpr (AMAX) {
                    // row maximum over this tile's columns: the BN4 lanes holding one row are contiguous in the wave
                    float bv = -INFINITY;
                    int bidx = 0x7fffffff;
                    if (m < p.M && n < p.N) {
                        float4 t4 = *reinterpret_cast<const float4*>(&ep[row * EP_LD + en4 * 4]);
                        t4.x = act_c<ACT>(fmaf(t4.x, sc4.x, bi4.x)); t4.y = act_c<ACT>(fmaf(t4.y, sc4.y, bi4.y));
                        t4.z = act_c<ACT>(fmaf(t4.z, sc4.z, bi4.z)); t4.w = act_c<ACT>(fmaf(t4.w, sc4.w, bi4.w));
                        bv = t4.x; bidx = n;
                        argmax_combine(bv, bidx, t4.y, n + 1);
                        argmax_combine(bv, bidx, t4.z, n + 2);
                        argmax_combine(bv, bidx, t4.w, n + 3);
                    }
#pragma unroll
                    for (int o2 = BN4 / 2; o2 > 0; o2 >>= 1) {
                        const float ov = __shfl_xor(bv, o2, 64);
                        const int oi = __shfl_xor(bidx, o2, 64);
                        argmax_combine(bv, bidx, ov, oi);
                    }
                    if (en4 == 0 && m < p.M) {
                        p.amax_val[(size_t)m * p.tiles_n + tile_n] = bv;
                        p.amax_idx[(size_t)m * p.tiles_n + tile_n] = bidx;
                    }
                    if (!y_out) continue;
                }
                if (m >= p.M || n >= p.N) continue;
                float4 v = *reinterpret_cast<const float4*>(&ep[row * EP_LD + en4 * 4]);
                if constexpr (STATS) stat_acc4(st_s1, st_s2, v, st_shift);
                v.x = fmaf(v.x, sc4.x, bi4.x); v.y = fmaf(v.y, sc4.y, bi4.y);
                v.z = fmaf(v.z, sc4.z, bi4.z); v.w = fmaf(v.w, sc4.w, bi4.w);
                const size_t o = (size_t)m * p.N + n;
                if (p.residual) {
                    if (p.res16) {
                        const uint2 rr = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.residual) + o);
                        v.x += __uint_as_float(rr.x << 16); v.y += __uint_as_float(rr.x & 0xffff0000u);
                        v.z += __uint_as_float(rr.y << 16); v.w += __uint_as_float(rr.y & 0xffff0000u);
                    } else {
                        const float4 rr = *reinterpret_cast<const float4*>(p.residual + o);
                        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                    }
                }
                v.x = act_c<ACT>(v.x); v.y = act_c<ACT>(v.y); v.z = act_c<ACT>(v.z); v.w = act_c<ACT>(v.w);
                if (p.y16) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(y_out) + o) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
                else *reinterpret_cast<float4*>(y_out + o) = v;
            }
            if (STATS && pass == NPASS - 1) {
                // column statistics of the result for the BatchNorm that follows: the row groups' shifted moments meet in
                // LDS (the tile image is dead once every thread has read its rows) and one thread per column adds them in order
                __syncthreads();
                float4* sc = reinterpret_cast<float4*>(ep);                       // [EP_RPI][BN4][2]
                sc[(erow * BN4 + en4) * 2] = st_s1;
                sc[(erow * BN4 + en4) * 2 + 1] = st_s2;
                __syncthreads();
                if (tid < BN && n0 + tid < p.N) {
                    const int q4 = tid >> 2, e = tid & 3;
                    float a1 = 0.f, a2 = 0.f;
#pragma unroll
                    for (int r = 0; r < EP_RPI; ++r) {
                        a1 += reinterpret_cast<const float*>(&sc[(r * BN4 + q4) * 2])[e];
                        a2 += reinterpret_cast<const float*>(&sc[(r * BN4 + q4) * 2 + 1])[e];
                    }
                    const float nrows = (float)min(BM, p.M - m0);
                    float psum, pm2;
                    stat_from_shifted(nrows, my_shift, a1, a2, &psum, &pm2);
                    p.stat_part[((size_t)tile_m * 2 + 0) * p.N + n0 + tid] = psum;
                    p.stat_part[((size_t)tile_m * 2 + 1) * p.N + n0 + tid] = pm2;
                }
            }
            __syncthreads();
        }
    });
}

template <int BM, int BN>
static void gb16_launch(GemmBf16Params& p, bool a16, hipStream_t st) {
    p.tiles_n = (p.N + BN - 1) / BN;
    gb16_set_order(p, a16 ? 2 : 4, 2);
    const dim3 grid((unsigned)((long long)((p.M + BM - 1) / BM) * p.tiles_n)), block(256);
    if (a16) {
        if (p.gate) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, true, false, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, false, false, true>), grid, block, 0, st, p);
    } else {
        if (p.gate) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, false>), grid, block, 0, st, p);
    }
}

// cfg: -1 = choose by shape, 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 128x256 tiles; x_bf16: x points to a bf16 [M][K] tensor.
int linear_bf16_f32(const void* xv, int x_bf16, const unsigned short* w, const float* scale, const float* bias, const float* residual,
                    const float* gate, int hw, float* y, int M, int K, int N, int act, int cfg, hipStream_t st) {
    const float* x = reinterpret_cast<const float*>(xv);
    if (!x || !w || !y || M <= 0 || K <= 0 || N <= 0 || (gate && hw <= 0)) { set_error("linear_bf16: bad arguments"); return ICAP_ERR_ARG; }
    if (K % 8 || N % 4) { set_error("linear_bf16: K=%d must be a multiple of 8 and N=%d of 4", K, N); return ICAP_ERR_ALIGN; }
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (scale && !aligned16(scale)) || (bias && !aligned16(bias)) ||
        (residual && !aligned16(residual)) || (gate && !aligned16(gate))) {
        set_error("linear_bf16: pointers must be 16-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    if ((long long)M * N > (1LL << 40) || (long long)((M + 63) / 64) * ((N + 63) / 64) > 0x7fffffffLL) {
        set_error("linear_bf16: problem too large");
        return ICAP_ERR_LIMIT;
    }
    GemmBf16Params p{x, w, scale, bias, residual, gate, y, M, N, K, gate ? hw : 1, act, 0, nullptr, nullptr};
    if (cfg < 0) {
        // the A tile is re-used across the tile's columns, so wide tiles cut the (fp32) activation traffic per flop
        const long long mt = (M + 127) / 128;
        if (N >= 256 && mt * ((N + 255) / 256) >= 256) cfg = 3;
        else if (N > 64 && mt * ((N + 127) / 128) >= 512) cfg = 0;
        else if (mt * ((N + 63) / 64) >= 512) cfg = 1;
        else cfg = 2;
    }
    switch (cfg) {
        case 0: gb16_launch<128, 128>(p, x_bf16 != 0, st); break;
        case 1: gb16_launch<128, 64>(p, x_bf16 != 0, st); break;
        case 2: gb16_launch<64, 64>(p, x_bf16 != 0, st); break;
        case 3: gb16_launch<128, 256>(p, x_bf16 != 0, st); break;
        default: set_error("linear_bf16: unknown tile configuration %d", cfg); return ICAP_ERR_ARG;
    }
    return launch_status("linear_bf16");
}

// KH x KW convolution, NHWC fp32 image, bf16 weights [Cout][KH][KW][Cin] (Cin % 32 == 0; or Cin == 4 with every weight
// row zero-padded from KH*KW*4 to a multiple of 32 values), fused scale / bias / residual / activation: the dense layers
// of the ResNet-18 and VGG-16 encoders in bf16 compute mode.
// cfg: -1 = choose by shape, 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 128x256, 4 = 256x128 tiles, 5 = the 256-row eight-wave
// kernel with direct-to-LDS loads (conv_bf16_wide.hip: bf16 x and y, bf16 or no residual, Cin % 64 == 0, Cout % 8 == 0).
// io_bf16: bit 0 = x is bf16 (Cin % 32 == 0 only), bit 1 = y is stored as bf16, bit 2 = residual is bf16.
int conv2d_bf16_wide_supported(int Cin, int Cout);
int conv2d_bf16_wide(const unsigned short* x, const unsigned short* w, const float* scale, const float* bias, const unsigned short* residual,
                     unsigned short* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho,
                     int Wo, int act, hipStream_t st);
constexpr long long CONV_WIDE_MIN_TILES = 192;   // 256-row tiles below which the narrower tile kernels fill the chip better

int conv2d_bf16_f32(const void* xv, const unsigned short* w, const float* scale, const float* bias, const void* resv, void* yv,
                    int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                    int io_bf16, int cfg, hipStream_t st) {
    const float* x = reinterpret_cast<const float*>(xv);
    const float* residual = reinterpret_cast<const float*>(resv);
    float* y = reinterpret_cast<float*>(yv);
    const bool x16 = (io_bf16 & 1) != 0;
    if (io_bf16 < 0 || io_bf16 > 7 || (x16 && Cin % 32)) { set_error("conv2d_bf16: bad io_bf16 flags %d for Cin=%d", io_bf16, Cin); return ICAP_ERR_ARG; }
    if (!x || !w || !y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad_t < 0 ||
        pad_l < 0 || Ho <= 0 || Wo <= 0) {
        set_error("conv2d_bf16: bad arguments");
        return ICAP_ERR_ARG;
    }
    if ((Ho - 1) * stride - pad_t >= H || (Wo - 1) * stride - pad_l >= W) {      // the last window must still touch the image
        set_error("conv2d_bf16: output size %dx%d does not fit input %dx%d", Ho, Wo, H, W);
        return ICAP_ERR_ARG;
    }
    if ((Cin % 32 && Cin != 4) || Cout % 4) {
        set_error("conv2d_bf16: Cin=%d must be 4 or a multiple of 32, and Cout=%d a multiple of 4", Cin, Cout);
        return ICAP_ERR_ALIGN;
    }
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (scale && !aligned16(scale)) || (bias && !aligned16(bias)) ||
        (residual && !aligned16(residual))) {
        set_error("conv2d_bf16: pointers must be 16-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    // Cin = 4: the weight rows hold KH * KW * 4 values padded with zeros to a multiple of 32
    const long long M = (long long)B * Ho * Wo, K = Cin == 4 ? ((long long)KH * KW * 4 + 31) / 32 * 32 : (long long)KH * KW * Cin;
    if (M > 0x7fffffffLL || K > 0x7fffffffLL || M * Cout > (1LL << 40) || ((M + 63) / 64) * ((Cout + 63) / 64) > 0x7fffffffLL) {
        set_error("conv2d_bf16: problem too large");
        return ICAP_ERR_LIMIT;
    }
    const bool wide_ok = x16 && (io_bf16 & 2) && (!residual || (io_bf16 & 4)) && conv2d_bf16_wide_supported(Cin, Cout);
    if (cfg < 0 && wide_ok) {
        const int bn = Cout > 128 ? 256 : (Cout > 64 ? 128 : 64);
        if (((M + 255) / 256) * ((Cout + bn - 1) / bn) >= CONV_WIDE_MIN_TILES && K >= 256) cfg = 5;    // at least four K tiles to pipeline
    }
    if (cfg == 5) {
        if (!wide_ok) { set_error("conv2d_bf16: the wide form takes bf16 x / y / residual, Cin %% 64 == 0, Cout %% 8 == 0"); return ICAP_ERR_ARG; }
        return conv2d_bf16_wide(reinterpret_cast<const unsigned short*>(xv), w, scale, bias, reinterpret_cast<const unsigned short*>(resv),
                                reinterpret_cast<unsigned short*>(yv), B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, act, st);
    }
    GemmBf16Params p{x, w, scale, bias, residual, nullptr, y, (int)M, Cout, (int)K, 1, act, 0, nullptr, nullptr,
                     H, W, Cin, KW, stride, pad_t, pad_l, Ho, Wo, KH * KW, (io_bf16 >> 1) & 1, (io_bf16 >> 2) & 1};
    if (cfg < 0) {
        const long long mt = (M + 127) / 128;
        if (Cout >= 256 && mt * ((Cout + 255) / 256) >= 256) cfg = 3;
        else if (Cout > 64 && mt * ((Cout + 127) / 128) >= 256) cfg = 0;
        else if (mt * ((Cout + 63) / 64) >= 512) cfg = 1;
        else cfg = 2;
    }
    auto launch = [&](auto bm, auto bn) {
        constexpr int BM = decltype(bm)::value, BN = decltype(bn)::value;
        p.tiles_n = (p.N + BN - 1) / BN;
        gb16_set_order(p, x16 ? 2 : 4, 2, KH * KW);
        const dim3 grid((unsigned)((long long)((p.M + BM - 1) / BM) * p.tiles_n)), block(256);
        if (Cin == 4) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, false, false, false, 2>), grid, block, 0, st, p);
        else if (x16) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, false, false, true, 1>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, false, false, false, 1>), grid, block, 0, st, p);
    };
    switch (cfg) {
        case 0: launch(std::integral_constant<int, 128>{}, std::integral_constant<int, 128>{}); break;
        case 1: launch(std::integral_constant<int, 128>{}, std::integral_constant<int, 64>{}); break;
        case 2: launch(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{}); break;
        case 3: launch(std::integral_constant<int, 128>{}, std::integral_constant<int, 256>{}); break;
        case 4: launch(std::integral_constant<int, 256>{}, std::integral_constant<int, 128>{}); break;
        default: set_error("conv2d_bf16: unknown tile configuration %d", cfg); return ICAP_ERR_ARG;
    }
    return launch_status("conv2d_bf16");
}

// Linear layer whose epilogue also emits, per output row and 128-column tile, the maximum and its column (first index on
// ties): part_val / part_idx [M][linear_bf16_argmax_parts(N)]; y (optional) [M][N].  The vocabulary projection of greedy
// decoding in bf16 compute mode.
int linear_bf16_argmax_parts(int N) { return (N + 127) / 128; }

int linear_bf16_argmax_f32(const float* x, const unsigned short* w, const float* scale, const float* bias, float* y, float* part_val,
                           int* part_idx, int M, int K, int N, hipStream_t st) {
    if (!x || !w || !part_val || !part_idx || M <= 0 || K <= 0 || N <= 0) { set_error("linear_bf16_argmax: bad arguments"); return ICAP_ERR_ARG; }
    if (K % 8 || N % 4) { set_error("linear_bf16_argmax: K=%d must be a multiple of 8 and N=%d of 4", K, N); return ICAP_ERR_ALIGN; }
    if (!aligned16(x) || !aligned16(w) || (y && !aligned16(y)) || (scale && !aligned16(scale)) || (bias && !aligned16(bias))) {
        set_error("linear_bf16_argmax: pointers must be 16-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    GemmBf16Params p{x, w, scale, bias, nullptr, nullptr, y, M, N, K, 1, ACT_NONE, (N + 127) / 128, part_val, part_idx};
    const long long tiles = (long long)((M + 127) / 128) * p.tiles_n;
    if (tiles > 0x7fffffffLL) { set_error("linear_bf16_argmax: problem too large"); return ICAP_ERR_LIMIT; }
    gb16_set_order(p, 4, 2);
    hipLaunchKernelGGL((gemm_bf16_kernel<128, 128, false, true>), dim3((unsigned)tiles), dim3(256), 0, st, p);
    return launch_status("linear_bf16_argmax");
}

// ---------------------------------------------------------------- fp32-accurate linear layers on split bf16 products
// w3: three bf16 planes [3][N][K] made by split3_f32 from the fp32 weights.  Dynamic LDS: the three single-buffered
// operand planes of a 128 x 128 x 32 tile take 60 KB, two workgroups per CU.
template <int BM, int BN>
constexpr int gb16_split_smem() {
    constexpr int stage = 3 * (BM + BN) * GB16_LDK * 2, ep = (BM < 64 ? BM : 64) * (BN + 4) * 4;
    return stage > ep ? stage : ep;
}

template <int BM, int BN, bool GATE, bool AMAX, bool A3 = false, bool STATS = false>
static int gb16_split_launch(GemmBf16Params& p, hipStream_t st) {
    static hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, GATE, AMAX, A3, 0, 3, STATS>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, gb16_split_smem<BM, BN>());
    if (attr_rc != hipSuccess) { set_error("linear_f32x3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr_rc)); return ICAP_ERR_HIP_BASE + (int)attr_rc; }
    p.tiles_n = (p.N + BN - 1) / BN;
    gb16_set_order(p, A3 ? 6 : 4, 6);
    const dim3 grid((unsigned)((long long)((p.M + BM - 1) / BM) * p.tiles_n)), block(256);
    constexpr int lds_bytes = gb16_split_smem<BM, BN>();
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, GATE, AMAX, A3, 0, 3, STATS>), grid, block, lds_bytes, st, p);
    return launch_status("linear_f32x3");
}

static int split_args_ok(const char* what, const float* x, const unsigned short* w3, const float* scale, const float* bias,
                         const float* residual, const float* gate, const float* y, int M, int K, int N) {
    if (!x || !w3 || M <= 0 || K <= 0 || N <= 0) { set_error("%s: bad arguments", what); return ICAP_ERR_ARG; }
    if (K % 8 || N % 4) { set_error("%s: K=%d must be a multiple of 8 and N=%d of 4", what, K, N); return ICAP_ERR_ALIGN; }
    if (!aligned16(x) || !aligned16(w3) || (y && !aligned16(y)) || (scale && !aligned16(scale)) || (bias && !aligned16(bias)) ||
        (residual && !aligned16(residual)) || (gate && !aligned16(gate)) || (((size_t)N * K * 2) & 15)) {
        set_error("%s: pointers (and every weight plane) must be 16-byte aligned", what);
        return ICAP_ERR_ALIGN;
    }
    if ((long long)M * N > (1LL << 40) || (long long)((M + 63) / 64) * ((N + 63) / 64) > 0x7fffffffLL) { set_error("%s: problem too large", what); return ICAP_ERR_LIMIT; }
    return ICAP_OK;
}

// the tall-activation kernel (gemm_x3t.hip)
int linear_f32x3_tall_supported(int K, int N, int hw, int gated);
int linear_f32x3_tall_f32(const float* x, const unsigned short* w3, const float* scale, const float* bias, const float* residual,
                          const float* gate, int hw, float* y, int M, int K, int N, hipStream_t st);

// cfg: -1 = choose by shape, 0 = 128x128, 1 = 64x128, 2 = 64x64 tiles, 3 = the tall-activation kernel (activations held in
// registers for the whole contraction; fp32 x, no activation function, K a multiple of 48: linear_f32x3_tall_supported)
int linear_f32x3_f32(const float* x, int x_split, const unsigned short* w3, const float* scale, const float* bias, const float* residual,
                     const float* gate, int hw, float* y, int M, int K, int N, int act, int cfg, hipStream_t st) {
    if (!y || (gate && hw <= 0) || (gate && x_split)) { set_error("linear_f32x3: bad arguments (pre-split activations take no gate)"); return ICAP_ERR_ARG; }
    if (x_split && (((size_t)M * K * 2) & 15)) { set_error("linear_f32x3: every activation plane must be 16-byte aligned"); return ICAP_ERR_ALIGN; }
    const int rc = split_args_ok("linear_f32x3", x, w3, scale, bias, residual, gate, y, M, K, N);
    if (rc != ICAP_OK) return rc;
    if (cfg == 3) {
        if (x_split || act != ACT_NONE) { set_error("linear_f32x3: the tall form takes fp32 activations and no activation function"); return ICAP_ERR_ARG; }
        return linear_f32x3_tall_f32(x, w3, scale, bias, residual, gate, hw, y, M, K, N, st);
    }
    GemmBf16Params p{x, w3, scale, bias, residual, gate, y, M, N, K, gate ? hw : 1, act, 0, nullptr, nullptr};
    if (cfg < 0) {
        const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
        const int waste = (N + 127) / 128 * 128 - N;            // columns of the last 128-wide tile that are padding
        cfg = (t128 >= 192 && waste * 4 <= N) ? 0 : ((long long)((M + 63) / 64) * ((N + 127) / 128) >= 192 && waste * 4 <= N ? 1 : 2);
    }
    if (x_split) {
        switch (cfg) {
            case 0: return gb16_split_launch<128, 128, false, false, true>(p, st);
            case 1: return gb16_split_launch<64, 128, false, false, true>(p, st);
            case 2: return gb16_split_launch<64, 64, false, false, true>(p, st);
            default: set_error("linear_f32x3: unknown tile configuration %d", cfg); return ICAP_ERR_ARG;
        }
    }
    switch (cfg) {
        case 0: return gate ? gb16_split_launch<128, 128, true, false>(p, st) : gb16_split_launch<128, 128, false, false>(p, st);
        case 1: return gate ? gb16_split_launch<64, 128, true, false>(p, st) : gb16_split_launch<64, 128, false, false>(p, st);
        case 2: return gate ? gb16_split_launch<64, 64, true, false>(p, st) : gb16_split_launch<64, 64, false, false>(p, st);
        default: set_error("linear_f32x3: unknown tile configuration %d", cfg); return ICAP_ERR_ARG;
    }
}

// ---- split-K form for layers with few output tiles and a long contraction (the decoder's 2048 -> 512 Linear at 1024 rows: 128
// tiles of 64 x 64 on 256 CUs): KS slices of the contraction run as KS times as many workgroups, each writing its partial
// products to ws [KS][M][N]; splitk_reduce_kernel adds the slices in order and applies the epilogue.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int ks, const float* __restrict__ scale,
                                                            const float* __restrict__ bias, const float* __restrict__ residual,
                                                            float* __restrict__ y, long long mn4, int n4, int act) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mn4) return;
    float4 v = reinterpret_cast<const float4*>(ws)[i];
    for (int z = 1; z < ks; ++z) {
        const float4 t = reinterpret_cast<const float4*>(ws)[(long long)z * mn4 + i];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const int c = (int)(i % n4);
    if (scale) { const float4 s4 = reinterpret_cast<const float4*>(scale)[c]; v.x *= s4.x; v.y *= s4.y; v.z *= s4.z; v.w *= s4.w; }
    if (bias) { const float4 b4 = reinterpret_cast<const float4*>(bias)[c]; v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
    if (residual) { const float4 r4 = reinterpret_cast<const float4*>(residual)[i]; v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
    if (act != ACT_NONE) { v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act); }
    reinterpret_cast<float4*>(y)[i] = v;
}

long long linear_f32x3_splitk_ws_floats(int M, int N, int ksplit) { return (long long)(ksplit > 1 ? ksplit : 0) * M * N; }

// cfg: 0 = 128x128, 1 = 64x128, 2 = 64x64 tiles; ksplit 2 .. 16; K must split into ksplit slices of whole 32-wide K steps
int linear_f32x3_splitk_f32(const float* x, const unsigned short* w3, const float* scale, const float* bias, const float* residual,
                            const float* gate, int hw, float* y, float* ws, long long ws_floats, int M, int K, int N, int act, int cfg,
                            int ksplit, hipStream_t st) {
    if (!y || !ws || (gate && hw <= 0) || ksplit < 2 || ksplit > 16) { set_error("linear_f32x3_splitk: bad arguments (ksplit 2 .. 16)"); return ICAP_ERR_ARG; }
    const int rc = split_args_ok("linear_f32x3_splitk", x, w3, scale, bias, residual, gate, y, M, K, N);
    if (rc != ICAP_OK) return rc;
    if (K % (ksplit * GB16_BK)) { set_error("linear_f32x3_splitk: K=%d does not split into %d slices of whole %d-wide steps", K, ksplit, GB16_BK); return ICAP_ERR_ARG; }
    if (!aligned16(ws) || ws_floats < linear_f32x3_splitk_ws_floats(M, N, ksplit)) {
        set_error("linear_f32x3_splitk: the workspace must be 16-byte aligned and hold %lld floats", linear_f32x3_splitk_ws_floats(M, N, ksplit));
        return ICAP_ERR_ARG;
    }
    GemmBf16Params p{x, w3, nullptr, nullptr, nullptr, gate, ws, M, N, K, gate ? hw : 1, ACT_NONE, 0, nullptr, nullptr};
    p.ksl = K / ksplit;
    auto launch = [&](auto bm, auto bn) -> int {
        constexpr int BM = decltype(bm)::value, BN = decltype(bn)::value;
        auto kern = gate ? gemm_bf16_kernel<BM, BN, true, false, false, 0, 3, false> : gemm_bf16_kernel<BM, BN, false, false, false, 0, 3, false>;
        static hipError_t rc_t = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, true, false, false, 0, 3, false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, gb16_split_smem<BM, BN>());
        static hipError_t rc_f = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, false, false, false, 0, 3, false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, gb16_split_smem<BM, BN>());
        if (rc_t != hipSuccess || rc_f != hipSuccess) { set_error("linear_f32x3_splitk: cannot raise the dynamic LDS limit"); return ICAP_ERR_HIP_BASE; }
        p.tiles_n = (p.N + BN - 1) / BN;
        gb16_set_order(p, 4, 6);
        const dim3 grid((unsigned)((long long)((p.M + BM - 1) / BM) * p.tiles_n), 1, (unsigned)ksplit), block(256);
        constexpr int lds_bytes = gb16_split_smem<BM, BN>();
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, st, p);
        return launch_status("linear_f32x3_splitk");
    };
    int lrc;
    switch (cfg) {
        case 0: lrc = launch(std::integral_constant<int, 128>{}, std::integral_constant<int, 128>{}); break;
        case 1: lrc = launch(std::integral_constant<int, 64>{}, std::integral_constant<int, 128>{}); break;
        case 2: lrc = launch(std::integral_constant<int, 64>{}, std::integral_constant<int, 64>{}); break;
        default: set_error("linear_f32x3_splitk: unknown tile configuration %d", cfg); return ICAP_ERR_ARG;
    }
    if (lrc != ICAP_OK) return lrc;
    const long long mn4 = (long long)M * N / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((mn4 + 255) / 256)), dim3(256), 0, st, ws, ksplit, scale, bias, residual, y, mn4, N / 4, act);
    return launch_status("linear_f32x3_splitk (reduce)");
}

// Plain split-product Linear (no scale, bias, residual, gate or activation) whose epilogue also leaves the column statistics
// of its result for a train-mode BatchNorm: stat_part [P][2][N] (sum, squared deviations), P = info[0] blocks of info[1] rows
// each (one per row tile).  cfg as in linear_f32x3_f32.
int linear_f32x3_stats_f32(const float* x, const unsigned short* w3, float* y, int M, int K, int N, int cfg, float* stat_part,
                           long long part_floats, int* info, hipStream_t st) {
    if (!y || !stat_part || !info) { set_error("linear_f32x3_stats: bad arguments"); return ICAP_ERR_ARG; }
    const int rc = split_args_ok("linear_f32x3_stats", x, w3, nullptr, nullptr, nullptr, nullptr, y, M, K, N);
    if (rc != ICAP_OK) return rc;
    GemmBf16Params p{x, w3, nullptr, nullptr, nullptr, nullptr, y, M, N, K, 1, ACT_NONE, 0, nullptr, nullptr};
    if (cfg < 0) {
        const long long t128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
        const int waste = (N + 127) / 128 * 128 - N;
        cfg = (t128 >= 192 && waste * 4 <= N) ? 0 : ((long long)((M + 63) / 64) * ((N + 127) / 128) >= 192 && waste * 4 <= N ? 1 : 2);
    }
    if (cfg < 0 || cfg > 2) { set_error("linear_f32x3_stats: unknown tile configuration %d", cfg); return ICAP_ERR_ARG; }
    const int bm = cfg == 0 ? 128 : 64;
    const long long parts = (long long)(M + bm - 1) / bm;            // one block of statistics per row tile
    if (parts * 2 * N > part_floats) {
        set_error("linear_f32x3_stats: %lld partials of %d columns need %lld floats, the buffer holds %lld", parts, N, parts * 2 * N, part_floats);
        return ICAP_ERR_LIMIT;
    }
    info[0] = (int)parts;
    info[1] = bm;
    p.stat_part = stat_part;
    switch (cfg) {
        case 0: return gb16_split_launch<128, 128, false, false, false, true>(p, st);
        case 1: return gb16_split_launch<64, 128, false, false, false, true>(p, st);
        default: return gb16_split_launch<64, 64, false, false, false, true>(p, st);
    }
}

// vocabulary projection with arg-max partials per 128-column tile (linear_bf16_argmax_parts(N) per row)
int linear_f32x3_argmax_f32(const float* x, int x_split, const unsigned short* w3, const float* scale, const float* bias, float* y,
                            float* part_val, int* part_idx, int M, int K, int N, hipStream_t st) {
    if (!part_val || !part_idx) { set_error("linear_f32x3_argmax: null partial buffers"); return ICAP_ERR_ARG; }
    if (x_split && (((size_t)M * K * 2) & 15)) { set_error("linear_f32x3_argmax: every activation plane must be 16-byte aligned"); return ICAP_ERR_ALIGN; }
    const int rc = split_args_ok("linear_f32x3_argmax", x, w3, scale, bias, nullptr, nullptr, y, M, K, N);
    if (rc != ICAP_OK) return rc;
    GemmBf16Params p{x, w3, scale, bias, nullptr, nullptr, y, M, N, K, 1, ACT_NONE, 0, part_val, part_idx};
    if (x_split) return M > 64 ? gb16_split_launch<128, 128, false, true, true>(p, st) : gb16_split_launch<64, 128, false, true, true>(p, st);
    return M > 64 ? gb16_split_launch<128, 128, false, true>(p, st) : gb16_split_launch<64, 128, false, true>(p, st);
}

// fp32 [n] -> three bf16 planes [3][n]: hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid); n a multiple of 2
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, long long n2) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256) {
        float a = src[2 * i], b = src[2 * i + 1];
        const unsigned h = pack_bf16(a, b);
        a -= __uint_as_float(h << 16); b -= __uint_as_float(h & 0xffff0000u);
        const unsigned m = pack_bf16(a, b);
        a -= __uint_as_float(m << 16); b -= __uint_as_float(m & 0xffff0000u);
        dst[i] = h;
        dst[n2 + i] = m;
        dst[2 * n2 + i] = pack_bf16(a, b);
    }
}

int split3_f32(const float* src, unsigned short* dst, long long n, hipStream_t st) {
    if (!src || !dst || n <= 0 || (n & 1)) { set_error("split3: bad arguments (n must be even)"); return ICAP_ERR_ARG; }
    long long blocks = (n / 2 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, reinterpret_cast<unsigned*>(dst), n / 2);
    return launch_status("split3");
}

// fp32 -> bf16 (round to nearest even), n a multiple of 2
__global__ __launch_bounds__(256) void to_bf16_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, long long n2) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long long)gridDim.x * 256)
        dst[i] = pack_bf16(src[2 * i], src[2 * i + 1]);
}

int to_bf16_f32(const float* src, unsigned short* dst, long long n, hipStream_t st) {
    if (!src || !dst || n <= 0 || (n & 1)) { set_error("to_bf16: bad arguments (n must be even)"); return ICAP_ERR_ARG; }
    long long blocks = (n / 2 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, reinterpret_cast<unsigned*>(dst), n / 2);
    return launch_status("to_bf16");
}

}  // namespace icap
