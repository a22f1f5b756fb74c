// fp32-accurate Linear layer / 1x1 convolution on the bf16 matrix cores for TALL, GATED activations - the MBConv projection
// convolutions at 28 x 28 / 14 x 14 / 7 x 7 (M = 50 k - 800 k pixels, K = 240 - 1152 expanded channels, N = 40 - 320):
//   y[M][N] = ((x[M][K] * gate[m / hw][k]) @ w[N][K]^T) * scale[n] + bias[n] (+ residual[m][n])
// (six exact bf16 products per multiply from three-way splits, see gemm_bf16_kernel<SPLIT = 3>).  The activations are read
// once and dominate the bytes; the tiled kernel stages them as three bf16 planes through LDS behind two workgroup barriers per
// 32-wide K slice.  Here a wave owns 32 rows for the whole contraction and nobody else ever sees its activations:
//   * per 48-wide K chunk the wave loads its [32][48] block with whole-line loads (one chunk ahead, in registers), drops it
//     into a WAVE-PRIVATE padded LDS block (no barrier: LDS operations of one wave complete in order) and reads it back as
//     lane (r, h) = 24 consecutive floats of row r - the A-fragment layout of v_mfma_f32_32x32x16_bf16 (which k a lane feeds
//     into a K step is free as long as the B fragment holds the same k: groups of 8 consecutive k of the lane's half chunk);
//     gate multiply and the exact hi / mid / lo split happen on those registers;
//   * the pre-split weights' K chunk [3][N][48] and the chunk of the tile's (at most four) gate rows are staged through
//     registers into one padded LDS buffer by all four waves, fetched one chunk ahead: two barriers per 48-wide chunk;
//   * 4 waves x 32 rows per workgroup, at most 256 registers and 71 KB of LDS: two workgroups per CU, so one's loads and
//     epilogue run under the other's matrix instructions.
// Every load is a plain load the compiler tracks (no direct-to-LDS loads: mixed with them, its waits fall back to "everything
// outstanding"), and every global access is 16 bytes per lane over whole lines: the texture path takes ~40 cycles per such
// instruction and ~120 for one whose lanes each start a line of their own, which is what a fragment-shaped load is.
#include "common.hpp"

namespace icap {

typedef __bf16 xt_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float xt_f32x2 __attribute__((ext_vector_type(2)));
typedef float xt_f32x4 __attribute__((ext_vector_type(4)));
typedef float xt_f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned xt_u32x4 __attribute__((ext_vector_type(4)));

struct X3tParams {
    const float* x;
    const unsigned short* w3;   // [3][N][K]
    const float* scale;
    const float* bias;
    const float* residual;
    const float* gate;          // [M / hw][K] or null
    float* y;
    int M, N, K, hw, tiles_n;
};

constexpr int XT_GI = 4;        // gate rows (images) a tile can touch: 128 rows of >= 43-pixel images
constexpr int XT_CK = 48;       // K chunk
constexpr int XT_XP = XT_CK + 4;   // floats per row of a wave's activation block: 13 16-byte units, odd

__device__ __forceinline__ unsigned xt_pack(float a, float b) {
    const xt_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, xt_bf16x2));
}

template <int RN>
constexpr int xt_lds_bytes() { return 3 * 32 * RN * (XT_CK + 8) * 2 + XT_GI * XT_CK * 4 + 4 * 32 * XT_XP * 4; }

// RN: column blocks of 32 per tile
template <int RN>
__global__ __launch_bounds__(256, 2) void gemm_x3t_kernel(const X3tParams p) {
    constexpr int CK = XT_CK;
    constexpr int HC = CK / 2;                // floats per lane and chunk
    constexpr int NG = HC / 8;                // MFMA K steps per chunk
    constexpr int NX = HC / 4;                // 16-byte pieces of x per lane and chunk
    constexpr int XU = CK / 4;                // 16-byte pieces per activation row and chunk
    constexpr int NP = 32 * RN;               // padded tile width
    constexpr int WPITCH = CK + 8;            // bf16 per weight row in LDS: 7 16-byte units, odd -> conflict-free fragment reads
    constexpr int WROWS = 3 * NP;             // plane-major rows of a weight chunk
    constexpr int WU = CK / 8;                // 16-byte units per weight row
    constexpr int NWL = (WROWS * WU + 255) / 256;   // 16-byte units of the weight chunk per thread
    constexpr int EP_LD = NP + 4;
    static_assert(WROWS * WPITCH * 2 + XT_GI * CK * 4 + 4 * 32 * XT_XP * 4 >= 128 * EP_LD * 4, "the result tile fits the staging buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char xt_smem[];
    unsigned short* wl = reinterpret_cast<unsigned short*>(xt_smem);                 // [3][NP][WPITCH]
    float* gl = reinterpret_cast<float*>(xt_smem + WROWS * WPITCH * 2);               // [XT_GI][CK] this chunk of the tile's gate rows
    float* xl = gl + XT_GI * CK;                                                      // [4 waves][32][XT_XP]
    float* ep = reinterpret_cast<float*>(xt_smem);                                    // result tile image, after the K loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int K = p.K, nchunk = K / CK;

    // XCD x walks a contiguous range of the tile list; neighbours in time share the activation panel
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tq = nwg >> 3, tr = nwg & 7;
    const int tlin = xcd * tq + (xcd < tr ? xcd : tr) + slot;
    const int tile_m = tlin / p.tiles_n, tile_n = tlin - tile_m * p.tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * NP;

    // ---- activations: piece q = lane + 64 i of the wave's [32 rows][XU pieces] block
    float* xw = xl + wave * 32 * XT_XP;
    unsigned x_off[NX];          // element offsets into x (M K < 2^31 is checked by the host)
    int x_dst[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int q = lane + 64 * i, row = q / XU, piece = q - row * XU;
        x_off[i] = (unsigned)min(m0 + wave * 32 + row, p.M - 1) * (unsigned)K + 4u * piece;
        x_dst[i] = row * XT_XP + 4 * piece;
    }
    const float* xf = xw + lr * XT_XP + lh * HC;         // this lane's fragment source
    // ---- gate rows: lane (r, h) multiplies by its row's image; thread t < XT_GI CK / 4 stages one 16-byte piece per chunk
    const bool gated = p.gate != nullptr;      // ungated: the staged gate rows stay at 1
    const int img0 = gated ? m0 / p.hw : 0;
    const float* gf = gl + (gated ? (min(m0 + wave * 32 + lr, p.M - 1) / p.hw - img0) * CK : 0) + lh * HC;
    const float* g_src = nullptr;
    if (gated && tid < XT_GI * XU) {
        const int which = tid / XU, piece = tid - which * XU;
        g_src = p.gate + (size_t)min(img0 + which, (p.M - 1) / p.hw) * K + 4 * piece;
    }
    // ---- weights: thread t owns 16-byte units t, t + 256, ... of the chunk image [3 NP rows][WU units]
    const unsigned short* w_src[NWL];
    int w_dst[NWL];
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
        const int u = min(tid + 256 * i, WROWS * WU - 1);
        const int row = u / WU, unit = u - row * WU;
        const int pl = row / NP, n = row - pl * NP;
        w_src[i] = p.w3 + ((size_t)pl * p.N + min(n0 + n, p.N - 1)) * K + unit * 8;
        w_dst[i] = row * WPITCH + unit * 8;
    }
    xt_u32x4 wr[NWL];
    xt_f32x4 gr = {1.f, 1.f, 1.f, 1.f};
    xt_f32x4 xr[NX];
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWL; ++i) wr[i] = *reinterpret_cast<const xt_u32x4*>(w_src[i] + c * CK);
        if (g_src) gr = *reinterpret_cast<const xt_f32x4*>(g_src + c * CK);
#pragma unroll
        for (int i = 0; i < NX; ++i) xr[i] = *reinterpret_cast<const xt_f32x4*>(p.x + x_off[i] + c * CK);
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            if (tid + 256 * i < WROWS * WU) *reinterpret_cast<xt_u32x4*>(wl + w_dst[i]) = wr[i];
        if (tid < XT_GI * XU) *reinterpret_cast<xt_f32x4*>(gl + 4 * tid) = gr;
#pragma unroll
        for (int i = 0; i < NX; ++i) *reinterpret_cast<xt_f32x4*>(xw + x_dst[i]) = xr[i];
    };

    xt_f32x16 acc[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    load_chunk(0);
    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();                       // nobody still reads the previous chunk's weights and gate rows
        store_chunk();
        load_chunk(min(c + 1, nchunk - 1));
        __syncthreads();
        // One basic block per chunk (no branch on the gate: ungated layers multiply by the ones the staging threads keep in the
        // gate rows), written as a software pipeline: the gate multiply and the exact three-way split of K step g + 1 are
        // independent of the matrix instructions of step g, so the scheduler can run them underneath.
        xt_bf16x8 af[3];
        auto split_step = [&](int g, xt_bf16x8 (&out)[3]) __attribute__((always_inline)) {
            const xt_f32x4 a0 = *reinterpret_cast<const xt_f32x4*>(xf + 8 * g), a1 = *reinterpret_cast<const xt_f32x4*>(xf + 8 * g + 4);
            const xt_f32x4 g0 = *reinterpret_cast<const xt_f32x4*>(gf + 8 * g), g1 = *reinterpret_cast<const xt_f32x4*>(gf + 8 * g + 4);
            const float v[8] = {a0[0] * g0[0], a0[1] * g0[1], a0[2] * g0[2], a0[3] * g0[3], a1[0] * g1[0], a1[1] * g1[1], a1[2] * g1[2], a1[3] * g1[3]};
            unsigned hi[4], mid[4], lo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[e] = xt_pack(v[2 * e], v[2 * e + 1]);
                const float r0 = v[2 * e] - __uint_as_float(hi[e] << 16), r1 = v[2 * e + 1] - __uint_as_float(hi[e] & 0xffff0000u);
                mid[e] = xt_pack(r0, r1);
                lo[e] = xt_pack(r0 - __uint_as_float(mid[e] << 16), r1 - __uint_as_float(mid[e] & 0xffff0000u));
            }
            out[0] = __builtin_bit_cast(xt_bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
            out[1] = __builtin_bit_cast(xt_bf16x8, make_uint4(mid[0], mid[1], mid[2], mid[3]));
            out[2] = __builtin_bit_cast(xt_bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
        };
        split_step(0, af);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // B fragments: lane (n, h) needs the same eight k of row n of each plane
            xt_bf16x8 bf[3][RN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    bf[pl][j] = *reinterpret_cast<const xt_bf16x8*>(wl + (pl * NP + j * 32 + lr) * WPITCH + lh * HC + 8 * g);
            xt_bf16x8 an[3] = {af[0], af[1], af[2]};
            if (g + 1 < NG) split_step(g + 1, an);
            // smallest terms first; consecutive MFMAs go to different accumulators
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < RN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa[q]], bf[pb[q]][j], acc[j], 0, 0, 0);
            af[0] = an[0]; af[1] = an[1]; af[2] = an[2];
        }
    }
    __syncthreads();                           // the staging buffers become the result tile

    // ---- epilogue, every wave on its own: its 32 x NP accumulator block -> a wave-private LDS image (row-major) -> 16-byte
    // pieces along the output rows; the residual pieces are requested before the image is written, and no barrier is needed
    constexpr int BN4 = NP / 4;                        // 16-byte pieces per tile row
    constexpr int NPC = (32 * BN4 + 63) / 64;          // pieces per lane
    float* ew = ep + wave * 32 * EP_LD;
    const int mw = m0 + wave * 32;
    xt_f32x4 rv[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int q = lane + 64 * i, row = q / BN4, n = n0 + 4 * (q - row * BN4);
        rv[i] = xt_f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.residual && row < 32 && mw + row < p.M && n < p.N) rv[i] = *reinterpret_cast<const xt_f32x4*>(p.residual + (size_t)(mw + row) * p.N + n);
    }
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) ew[((e & 3) + 8 * (e >> 2) + 4 * lh) * EP_LD + j * 32 + lr] = acc[j][e];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int q = lane + 64 * i, row = q / BN4, c4 = q - row * BN4, n = n0 + 4 * c4;
        if (row >= 32 || mw + row >= p.M || n >= p.N) continue;
        xt_f32x4 v = *reinterpret_cast<const xt_f32x4*>(ew + row * EP_LD + 4 * c4);
        xt_f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (p.scale) sc = *reinterpret_cast<const xt_f32x4*>(p.scale + n);
        if (p.bias) bi = *reinterpret_cast<const xt_f32x4*>(p.bias + n);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = fmaf(v[t], sc[t], bi[t]) + rv[i][t];
        *reinterpret_cast<xt_f32x4*>(p.y + (size_t)(mw + row) * p.N + n) = v;
    }
}

template <int RN>
static int xt_launch(X3tParams& p, hipStream_t st) {
    constexpr int NP = 32 * RN;
    static hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3t_kernel<RN>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, xt_lds_bytes<RN>());
    if (attr_rc != hipSuccess) { set_error("linear_f32x3 (tall): cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr_rc)); return ICAP_ERR_HIP_BASE + (int)attr_rc; }
    p.tiles_n = (p.N + NP - 1) / NP;
    const long long wgs = (long long)((p.M + 127) / 128) * p.tiles_n;
    if (wgs > 0x7fffffffLL) { set_error("linear_f32x3 (tall): problem too large"); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL((gemm_x3t_kernel<RN>), dim3((unsigned)wgs), dim3(256), xt_lds_bytes<RN>(), st, p);
    return launch_status("linear_f32x3 (tall)");
}

// 1 when the tall form has a kernel for this contraction length and width
int linear_f32x3_tall_supported(int K, int N, int hw, int gated) {
    if (N % 4 || N <= 0 || K <= 0 || K % XT_CK) return 0;
    if (gated && hw < 43) return 0;            // a 128-row tile must not touch more than XT_GI images
    return 1;
}

// the tall-activation form of linear_f32x3_f32 (its cfg 3): x fp32 [M][K]
int linear_f32x3_tall_f32(const float* x, const unsigned short* w3, const float* scale, const float* bias, const float* residual,
                          const float* gate, int hw, float* y, int M, int K, int N, hipStream_t st) {
    if (!linear_f32x3_tall_supported(K, N, hw, gate != nullptr)) {
        set_error("linear_f32x3 (tall): K=%d must be a multiple of %d, N=%d of 4, gated images at least 43 pixels", K, XT_CK, N);
        return ICAP_ERR_ARG;
    }
    if ((long long)M * K >= (1LL << 31)) { set_error("linear_f32x3 (tall): M K = %lld elements, the limit is 2^31", (long long)M * K); return ICAP_ERR_LIMIT; }
    X3tParams p{x, w3, scale, bias, residual, gate, y, M, N, K, gate ? hw : 1, 0};
    // column blocks per tile: the whole width up to 4 blocks (the register budget of two workgroups per CU), else the
    // split into equal tiles that pads least, wider tiles first (every column tile re-reads the activations)
    const int blocks = (N + 31) / 32;
    int rn = blocks;
    if (blocks > 4) {
        int best_pad = 1 << 30;
        for (int r = 4; r >= 2; --r) {
            const int pad = (blocks + r - 1) / r * r - blocks;
            if (pad < best_pad) { rn = r; best_pad = pad; }
        }
    }
    switch (rn) {
        case 1: return xt_launch<1>(p, st);
        case 2: return xt_launch<2>(p, st);
        case 3: return xt_launch<3>(p, st);
        default: return xt_launch<4>(p, st);
    }
}

}  // namespace icap
