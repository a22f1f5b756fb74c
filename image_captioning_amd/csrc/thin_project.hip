// 1x1 "projection" convolutions of the high-resolution MBConv blocks (EfficientNet-B0: 32 -> 16 at 112 x 112, 96 / 144 -> 24
// at 56 x 56) as a STREAMING kernel:
//   y[M][N] = ((x[M][K] * gate[m / hw][k]) @ w[N][K]^T) * scale[n] + bias[n] (+ residual[m][n]),   N <= 32, K in {32, 96, 144}
// These layers read 0.8 - 1.6 GB of activations at B = 1024 for a few GFLOP: pure HBM streams.  A tiled GEMM (operand tiles
// through LDS, barriers per K slice) keeps too few bytes in flight per CU for that.  Here
//   * the whole weight matrix (<= 62 KB fp32) sits in LDS for the workgroup's lifetime (row pitch K + 4 floats: the B fragment
//     reads of 32 consecutive rows fall in distinct bank groups);
//   * a wave owns 32 rows at a time - one contiguous run of 32 K floats - and fetches them with whole-cache-line loads
//     (lane l takes the l-th float4), parks them in a wave-private padded LDS image and reads them back in the fp32 MFMA's A
//     layout: lane (r, h) takes the contiguous half row k in [h K/2, (h + 1) K/2), which v_mfma_f32_32x32x2_f32 is free to
//     use as "its" k of every step, because the contraction order is arbitrary as long as the B fragment uses the same
//     assignment (loading that layout straight from memory - 64 different lines per instruction - measured 10-15 % slower);
//   * there is NO barrier and no hand-counted wait in the loop: waves are independent, two or more per SIMD, each with the
//     NEXT row block (4 - 18 KB, plus its residual rows) in flight while the current one is multiplied;
//   * the squeeze-excite gate multiplies the registers (the block's one or two gate rows are staged in wave-private LDS);
//   * the 32 x N result goes back through the same wave-private LDS area so that the stores are 16 bytes per lane along the row, with the
//     folded BatchNorm and the residual applied there.
// Products are exact fp32 (the fp32 matrix instruction), so the kernel serves the plain and the "f32x3" modes alike.
#include "common.hpp"

namespace icap {

typedef float tp_f32x16 __attribute__((ext_vector_type(16)));
typedef float tp_f32x4 __attribute__((ext_vector_type(4)));     // register sets are arrays of the NATIVE vector type: arrays of
                                                                 // HIP's float4 struct passed by reference end up in scratch memory

struct ThinParams {
    const float* x;
    const float* w;
    const float* scale;
    const float* bias;
    const float* residual;
    const float* gate;
    float* y;
    long long M;
    int N, hw, blocks_per_wave;
    long long nblocks;
};

// WPB: waves per workgroup (they share the LDS copy of the weights)
template <int K, int NCB, int WPB>
__global__ __launch_bounds__(64 * WPB) void thin_project_kernel(const ThinParams p) {
    constexpr int KH = K / 2, K4 = K / 4, WP = K + 4, NP = 32 * NCB, TP = NP + 4;
    constexpr int NL = K / 8;                             // 16-byte loads per lane and 32-row block (32 K floats / 64 lanes / 4)
    static_assert(K % 8 == 0 && K4 % 2 == 0, "half rows are whole float4s; the padded pitch must be an odd number of 16-byte units");
    static_assert(32 * WP >= 32 * TP, "the result tile re-uses the block's LDS image");
    extern __shared__ __attribute__((aligned(16))) float tp_lds[];
    float* wl = tp_lds;                                   // [NP][WP] weights, rows >= N zero
    float* gl = wl + NP * WP;                             // [WPB waves][2][K] gate rows of the wave's current block
    float* xl = gl + WPB * 2 * K;                         // [WPB waves][32][WP] the wave's current rows, then its result tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;

    for (int i = tid; i < NP * K4; i += 64 * WPB) {
        const int n = i / K4, k4 = i - n * K4;
        const float4 v = n < p.N ? *reinterpret_cast<const float4*>(p.w + (size_t)n * K + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(wl + n * WP + 4 * k4) = v;
    }
    __syncthreads();

    float* gw = gl + wave * 2 * K;
    float* xw = xl + wave * 32 * WP;
    // this wave's contiguous range of 32-row blocks (neighbouring blocks share an image, hence a gate row)
    const long long gwave = (long long)blockIdx.x * WPB + wave;
    const long long b0 = gwave * p.blocks_per_wave;
    const long long b1 = b0 + p.blocks_per_wave < p.nblocks ? b0 + p.blocks_per_wave : p.nblocks;
    int staged_img = -1;
    // epilogue coordinates: thread covers columns 4 c4 .. + 3 of rows er, er + RS, ...
    constexpr int C4 = NP / 4, RS = 64 / C4, NR = 32 / RS;
    const int c4 = lane % C4, er = lane / C4;
    const int n0 = 4 * c4;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), bi4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 < p.N) {
        if (p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + n0);
        if (p.bias) bi4 = *reinterpret_cast<const float4*>(p.bias + n0);
    }
    // a block's rows are one contiguous run of 32 K floats: lane l of load i takes float4 number 64 i + l of it - whole cache
    // lines per instruction - and files it under (row, column) = divmod(64 i + l, K / 4) in the padded LDS image
    int xoff[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int e = 64 * i + lane;
        xoff[i] = (e / K4) * WP + 4 * (e % K4);
    }
    const long long total4 = p.M * (long long)K4;         // float4s in x

    // Two register sets: the next block's rows (and residual rows) are requested before the current block is multiplied.
    auto load_block = [&](long long blk, tp_f32x4 (&xr)[NL], tp_f32x4 (&rr)[NR]) __attribute__((always_inline)) {
        const int m0 = (int)blk * 32;                     // M < 2^31 (checked by the host)
        const tp_f32x4* src = reinterpret_cast<const tp_f32x4*>(p.x) + (size_t)m0 * K4;
        const long long room = total4 - (long long)m0 * K4;                 // float4s from the block's start to the end of x
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = 64 * i + lane;
            xr[i] = __builtin_nontemporal_load(src + (e < room ? e : 0));   // read once: streaming; rows past M: any valid address (their results are not stored)
        }
        if (p.residual) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int m = min(m0 + er + i * RS, (int)p.M - 1);
                rr[i] = *reinterpret_cast<const tp_f32x4*>(p.residual + (size_t)m * p.N + min(n0, p.N - 4));
            }
        }
    };
    auto compute_block = [&](long long blk, const tp_f32x4 (&xr)[NL], const tp_f32x4 (&rr)[NR]) __attribute__((always_inline)) {
        const int m0 = (int)blk * 32;
#pragma unroll
        for (int i = 0; i < NL; ++i) *reinterpret_cast<tp_f32x4*>(xw + xoff[i]) = xr[i];
        const float* g = nullptr;
        if (p.gate) {
            // gate rows of the (at most two) images this block touches, staged once per image change
            const int row = m0 + lr < (int)p.M ? m0 + lr : (int)p.M - 1;
            const int img0 = m0 / p.hw;                   // hw >= 32: a block touches at most two images
            if (img0 != staged_img) {
                const int last = ((int)p.M - 1) / p.hw;
                for (int i = lane; i < 2 * K4; i += 64) {
                    const int which = i / K4, k4 = i - which * K4;
                    const int img = img0 + which < last ? img0 + which : last;
                    *reinterpret_cast<float4*>(gw + which * K + 4 * k4) = *reinterpret_cast<const float4*>(p.gate + (size_t)img * K + 4 * k4);
                }
                staged_img = img0;
            }
            g = gw + (row >= (img0 + 1) * p.hw ? K : 0) + lh * KH;
        }
        // wave-private LDS: the wave's own writes are visible to its reads once they have completed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tp_f32x16 acc[NCB];
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        const float* xrow = xw + lr * WP + lh * KH;       // lane (r, h): the half row k in [h K/2, (h + 1) K/2) of row r
#pragma unroll
        for (int k = 0; k < KH; k += 4) {
            float4 xv = *reinterpret_cast<const float4*>(xrow + k);
            if (g) {
                const float4 gv = *reinterpret_cast<const float4*>(g + k);
                xv.x *= gv.x; xv.y *= gv.y; xv.z *= gv.z; xv.w *= gv.w;
            }
            float4 wv[NCB];
#pragma unroll
            for (int j = 0; j < NCB; ++j) wv[j] = *reinterpret_cast<const float4*>(wl + (j * 32 + lr) * WP + lh * KH + k);
#pragma unroll
            for (int j = 0; j < NCB; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.x, wv[j].x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.y, wv[j].y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.z, wv[j].z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.w, wv[j].w, acc[j], 0, 0, 0);
            }
        }
        // ---- result tile (over the block's rows, which every lane has finished reading) -> rows of 16-byte stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) xw[((e & 3) + 8 * (e >> 2) + 4 * lh) * TP + j * 32 + lr] = acc[j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (n0 < p.N) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = er + i * RS;
                const int m = m0 + r;
                if (m < (int)p.M) {
                    float4 v = *reinterpret_cast<const float4*>(xw + r * TP + n0);
                    v.x = fmaf(v.x, sc4.x, bi4.x); v.y = fmaf(v.y, sc4.y, bi4.y);
                    v.z = fmaf(v.z, sc4.z, bi4.z); v.w = fmaf(v.w, sc4.w, bi4.w);
                    if (p.residual) { v.x += rr[i][0]; v.y += rr[i][1]; v.z += rr[i][2]; v.w += rr[i][3]; }
                    *reinterpret_cast<float4*>(p.y + (size_t)m * p.N + n0) = v;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tile is read before the next block's rows overwrite it
    };
    // every load is unconditional (past the range the last block is fetched again): a register set that is only sometimes
    // written is kept in scratch memory by the compiler
    if (b0 >= b1) return;
    const long long bl = b1 - 1;
    tp_f32x4 xa0[NL], xa1[NL];
    tp_f32x4 rr0[NR], rr1[NR];
    load_block(b0, xa0, rr0);
    for (long long blk = b0; blk < b1; blk += 2) {
        load_block(blk + 1 < b1 ? blk + 1 : bl, xa1, rr1);
        compute_block(blk, xa0, rr0);
        load_block(blk + 2 < b1 ? blk + 2 : bl, xa0, rr0);
        if (blk + 1 < b1) compute_block(blk + 1, xa1, rr1);
    }
}

template <int K, int NCB, int WPB>
static int thin_launch(ThinParams& p, hipStream_t st) {
    constexpr int NP = 32 * NCB;
    constexpr int lds_bytes = (NP * (K + 4) + WPB * 2 * K + WPB * 32 * (K + 4)) * 4;
    static hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_project_kernel<K, NCB, WPB>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (attr_rc != hipSuccess) { set_error("linear_thin: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr_rc)); return ICAP_ERR_HIP_BASE + (int)attr_rc; }
    p.nblocks = (p.M + 31) / 32;
    // about eight waves per SIMD's worth of workgroups, each wave a contiguous run of blocks
    const long long want_waves = 256LL * 4 * 8;
    long long bpw = (p.nblocks + want_waves - 1) / want_waves;
    if (bpw < 1) bpw = 1;
    if (bpw > 0x7fffffffLL) { set_error("linear_thin: problem too large"); return ICAP_ERR_LIMIT; }
    p.blocks_per_wave = (int)bpw;
    const long long waves = (p.nblocks + bpw - 1) / bpw;
    const long long wgs = (waves + WPB - 1) / WPB;
    if (wgs > 0x7fffffffLL) { set_error("linear_thin: problem too large"); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL((thin_project_kernel<K, NCB, WPB>), dim3((unsigned)wgs), dim3(64 * WPB), lds_bytes, st, p);
    return launch_status("linear_thin");
}

// 1 when linear_thin_f32 has a kernel for this contraction length and width
int linear_thin_supported(int K, int N) {
    if (N <= 0 || N > 32 || N % 4) return 0;
    return K == 32 || K == 96 || K == 144;
}

int linear_thin_f32(const float* x, const float* w, const float* scale, const float* bias, const float* residual, const float* gate,
                    int hw, float* y, long long M, int K, int N, hipStream_t st) {
    if (!x || !w || !y || M <= 0 || M > 0x7fffffffLL - 64 || (gate && hw < 32)) {
        set_error("linear_thin: bad arguments (M < 2^31; a gate needs hw >= 32 rows per image)");
        return ICAP_ERR_ARG;
    }
    if (!linear_thin_supported(K, N)) { set_error("linear_thin: no kernel for K=%d N=%d (K in {32, 96, 144}, N <= 32, a multiple of 4)", K, N); return ICAP_ERR_ARG; }
    if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (scale && !aligned16(scale)) || (bias && !aligned16(bias)) ||
        (residual && !aligned16(residual)) || (gate && !aligned16(gate))) {
        set_error("linear_thin: pointers must be 16-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    ThinParams p{x, w, scale, bias, residual, gate, y, M, N, gate ? hw : 1, 0, 0};
    switch (K) {
        case 32: return thin_launch<32, 1, 4>(p, st);
        case 96: return thin_launch<96, 1, 8>(p, st);
        default: return thin_launch<144, 1, 4>(p, st);       // one workgroup per CU (99 KB of LDS), one wave per SIMD: two 72-register sets
    }
}

}  // namespace icap
