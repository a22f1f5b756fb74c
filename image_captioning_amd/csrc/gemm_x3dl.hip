// fp32-accurate Linear layer on the bf16 matrix cores with BOTH operands pre-split and streamed straight into LDS:
//   y[M][N] = act((x[M][K] @ w[N][K]^T) * scale[n] + bias[n] (+ residual[m][n]))
// x3 [3][M][K] and w3 [3][N][K] are the exact three-way bf16 splits of the fp32 operands (split3_f32: hi, mid, lo planes),
// six v_mfma_f32_32x32x16_bf16 per multiply as in gemm_bf16_kernel<SPLIT = 3>.  What is different here is the staging:
// no operand byte passes through a register.  Every lane issues global_load_lds_dwordx4 (16 bytes global -> LDS, a wave
// fills 1 KB of consecutive LDS), three K slices of 32 are in flight in a ring of LDS stages, and the only waits in the K
// loop are one s_waitcnt vmcnt for the oldest stage and one s_barrier.  The batched greedy decoder is the user: its
// activations are tiny against the weights (M = images, K = 512 .. 2048), so splitting them once in the producing kernel's
// epilogue (y3) or in one elementwise pass costs nothing, and the decoder GEMMs were the launches furthest from the
// matrix-core roofline (register staging exposed one memory latency per K step at one or two workgroups per CU).
//
// LDS layout of a stage: [A hi | A mid | A lo | W hi | W mid | W lo], a plane = rows of 32 bf16 (64 bytes, four 16-byte
// chunks), chunk c of row r stored at position c ^ ((r >> 2) & 3): a wave-wide direct load can only write consecutive
// 16-byte slots, so the swizzle is applied to the GLOBAL address each lane fetches (lane l of a wave-load owns slot
// (row l / 4, position l % 4) and fetches chunk (l % 4) ^ ((l >> 4) & 3) of that row); the MFMA fragment reads (lane
// (r, h) needs chunk 2 s + h of row r) then hit 16 distinct 16-byte bank groups per 16 lanes.
#include "common.hpp"

namespace icap {

typedef __bf16 xd_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xd_bf16x2 __attribute__((ext_vector_type(2)));
typedef float xd_f32x2 __attribute__((ext_vector_type(2)));
typedef float xd_f32x16 __attribute__((ext_vector_type(16)));

struct X3dlParams {
    const unsigned short* x3;   // [3][M][K]
    const unsigned short* w3;   // [3][N][K]
    const float* scale;
    const float* bias;
    const float* residual;      // fp32 [M][N] or null
    float* y;                   // fp32 [M][N] or null
    unsigned short* y3;         // or / and the result as three bf16 planes [3][M][N]: the next GEMM's pre-split input
    int M, N, K, act, tiles_n, m_fast;
    float* amax_val;            // optional [M][tiles_n]: per-tile row maxima (arg-max partials), BN = 128 only
    int* amax_idx;
};

constexpr int XD_BK = 32;        // K elements per stage
constexpr int XD_STAGES = 3;     // LDS ring: two stages in flight while one is multiplied

// 16 bytes per lane, global -> LDS: lane l of the wave writes lds_wave_base + 16 l (lds_wave_base is the same in every lane)
__device__ __forceinline__ void xd_load16(const unsigned short* g, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ unsigned xd_pack(float a, float b) {
    const xd_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, xd_bf16x2));
}

template <int BM, int BN>
constexpr int xd_stage_bytes() { return 3 * (BM + BN) * XD_BK * 2; }

template <int BM, int BN>
constexpr int xd_smem_bytes() {
    constexpr int ring = XD_STAGES * xd_stage_bytes<BM, BN>(), ep = (BM < 64 ? BM : 64) * (BN + 4) * 4;
    return ring > ep ? ring : ep;
}

template <int BM, int BN, bool AMAX, int MINWG>
__global__ __launch_bounds__(256, MINWG) void gemm_x3dl_kernel(const X3dlParams p) {
    static_assert(BM % 64 == 0 && BN % 64 == 0, "a wave-wide direct load fills 16 rows; four waves fill 64");
    constexpr int TM = BM / 2, TN = BN / 2, RM = TM / 32, RN = TN / 32;
    constexpr int APB = BM * XD_BK * 2, BPB = BN * XD_BK * 2;       // bytes per operand plane
    constexpr int STAGE = xd_stage_bytes<BM, BN>();
    constexpr int EP_LD = BN + 4, EP_ROWS = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char xd_smem[];
    float* ep = reinterpret_cast<float*>(xd_smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    // XCD x walks a contiguous range of the tile list, neighbours in time share the larger operand's panel
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

    // ---- direct loads: this lane's slot in a wave-load is (row wave * 16 + lane / 4 of a 64-row group, position lane % 4);
    // rows past the matrix are clamped (their products land in result rows / columns that are never stored)
    const int srow = wave * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const size_t a_plane = (size_t)p.M * p.K, b_plane = (size_t)p.N * p.K;
    const unsigned short* a_src[BM / 64];
    const unsigned short* b_src[BN / 64];
#pragma unroll
    for (int h = 0; h < BM / 64; ++h) a_src[h] = p.x3 + (size_t)min(m0 + h * 64 + srow, p.M - 1) * p.K + chunk * 8;
#pragma unroll
    for (int h = 0; h < BN / 64; ++h) b_src[h] = p.w3 + (size_t)min(n0 + h * 64 + srow, p.N - 1) * p.K + chunk * 8;
    constexpr int LOADS = 3 * (BM / 64 + BN / 64);      // direct loads per lane and stage
    auto issue = [&](int t, int buf) {
        unsigned char* base = xd_smem + buf * STAGE + wave * 1024;
        const int k0 = t * XD_BK;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int h = 0; h < BM / 64; ++h)
                xd_load16(a_src[h] + pl * a_plane + k0, base + pl * APB + h * 4096);
#pragma unroll
            for (int h = 0; h < BN / 64; ++h)
                xd_load16(b_src[h] + pl * b_plane + k0, base + 3 * APB + pl * BPB + h * 4096);
        }
    };

    xd_f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addresses inside a plane: row r, chunk q at r * 64 + ((q ^ ((r >> 2) & 3)) * 16)
    const int sw = (lr >> 2) & 3;
    auto compute = [&](int buf) {
        const unsigned char* sa = xd_smem + buf * STAGE;
        const unsigned char* sb = sa + 3 * APB;
#pragma unroll
        for (int s = 0; s < XD_BK / 16; ++s) {
            const int off = ((2 * s + lh) ^ sw) * 16;
            xd_bf16x8 af[3][RM], bf[3][RN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < RM; ++i)
                    af[pl][i] = *reinterpret_cast<const xd_bf16x8*>(sa + pl * APB + (wm * TM + i * 32 + lr) * 64 + off);
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    bf[pl][j] = *reinterpret_cast<const xd_bf16x8*>(sb + pl * BPB + (wn * TN + j * 32 + lr) * 64 + off);
            }
            // smallest terms first; consecutive MFMAs go to different accumulators
            constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < RM; ++i)
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa[q]][i], bf[pb[q]][j], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = p.K / XD_BK;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    for (int t = 0; t < nk; ++t) {
        // stage t has landed once at most the next stage's loads are outstanding; the barrier makes every wave's part of it
        // visible and says that everybody is done reading the stage the new loads overwrite
        if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 2 < nk) issue(t + 2, (t + 2) % XD_STAGES);
        compute(t % XD_STAGES);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // every wave is done with the ring: the epilogue's tile image overwrites it
    asm volatile("" ::: "memory");

    // ---- epilogue: accumulators -> LDS (row-major) -> 16-byte stores along the output row ----
    constexpr int BN4 = BN / 4;
    constexpr int EP_RPI = 256 / BN4;
    constexpr int NPASS = BM / EP_ROWS;
    const int en4 = tid % BN4, erow = tid / BN4;
    const int n = n0 + en4 * 4;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), bi4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < p.N) {
        if (p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + n);
        if (p.bias) bi4 = *reinterpret_cast<const float4*>(p.bias + n);
    }
    const size_t y_plane = (size_t)p.M * p.N;
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
                    if (!p.y && !p.y3) continue;
                }
                if (m >= p.M || n >= p.N) continue;
                float4 v = *reinterpret_cast<const float4*>(&ep[row * EP_LD + en4 * 4]);
                v.x = fmaf(v.x, sc4.x, bi4.x); v.y = fmaf(v.y, sc4.y, bi4.y);
                v.z = fmaf(v.z, sc4.z, bi4.z); v.w = fmaf(v.w, sc4.w, bi4.w);
                const size_t o = (size_t)m * p.N + n;
                if (p.residual) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.residual + o);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                v.x = act_c<ACT>(v.x); v.y = act_c<ACT>(v.y); v.z = act_c<ACT>(v.z); v.w = act_c<ACT>(v.w);
                if (p.y) *reinterpret_cast<float4*>(p.y + o) = v;
                if (p.y3) {
                    // the result as its exact three-way bf16 split (what split3_f32 would make of y)
                    const uint2 hi = make_uint2(xd_pack(v.x, v.y), xd_pack(v.z, v.w));
                    float4 r = make_float4(v.x - __uint_as_float(hi.x << 16), v.y - __uint_as_float(hi.x & 0xffff0000u),
                                           v.z - __uint_as_float(hi.y << 16), v.w - __uint_as_float(hi.y & 0xffff0000u));
                    const uint2 mid = make_uint2(xd_pack(r.x, r.y), xd_pack(r.z, r.w));
                    r = make_float4(r.x - __uint_as_float(mid.x << 16), r.y - __uint_as_float(mid.x & 0xffff0000u),
                                    r.z - __uint_as_float(mid.y << 16), r.w - __uint_as_float(mid.y & 0xffff0000u));
                    *reinterpret_cast<uint2*>(p.y3 + o) = hi;
                    *reinterpret_cast<uint2*>(p.y3 + y_plane + o) = mid;
                    *reinterpret_cast<uint2*>(p.y3 + 2 * y_plane + o) = make_uint2(xd_pack(r.x, r.y), xd_pack(r.z, r.w));
                }
            }
            __syncthreads();
        }
    });
}

template <int BM, int BN, bool AMAX>
static int xd_launch(X3dlParams& p, hipStream_t st) {
    constexpr int lds_bytes = xd_smem_bytes<BM, BN>();
    constexpr int MINWG = lds_bytes <= 80 * 1024 ? 2 : 1;        // workgroups per CU the ring leaves room for
    static hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3dl_kernel<BM, BN, AMAX, MINWG>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (attr_rc != hipSuccess) { set_error("linear_x3dl: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr_rc)); return ICAP_ERR_HIP_BASE + (int)attr_rc; }
    p.tiles_n = (p.N + BN - 1) / BN;
    p.m_fast = (long long)p.N > (long long)p.M;          // both operands are 6 bytes per element: the larger matrix is the one with more rows
    const dim3 grid((unsigned)((long long)((p.M + BM - 1) / BM) * p.tiles_n)), block(256);
    hipLaunchKernelGGL((gemm_x3dl_kernel<BM, BN, AMAX, MINWG>), grid, block, lds_bytes, st, p);
    return launch_status("linear_x3dl");
}

static int xd_args_ok(const char* what, const unsigned short* x3, const unsigned short* w3, const float* scale, const float* bias,
                      const float* residual, const float* y, const unsigned short* y3, int M, int K, int N) {
    if (!x3 || !w3 || M <= 0 || K <= 0 || N <= 0) { set_error("%s: bad arguments", what); return ICAP_ERR_ARG; }
    if (K % XD_BK || N % 4) { set_error("%s: K=%d must be a multiple of %d and N=%d of 4", what, K, XD_BK, N); return ICAP_ERR_ALIGN; }
    if (!aligned16(x3) || !aligned16(w3) || (y && !aligned16(y)) || (y3 && (reinterpret_cast<uintptr_t>(y3) & 7u)) || (scale && !aligned16(scale)) ||
        (bias && !aligned16(bias)) || (residual && !aligned16(residual)) || (((size_t)M * K * 2) & 15) || (((size_t)N * K * 2) & 15) ||
        (y3 && (((size_t)M * N * 2) & 7))) {
        set_error("%s: pointers (and every operand plane) must be 16-byte aligned", what);
        return ICAP_ERR_ALIGN;
    }
    if ((long long)M * N > (1LL << 40) || (long long)((M + 63) / 64) * ((N + 63) / 64) > 0x7fffffffLL) { set_error("%s: problem too large", what); return ICAP_ERR_LIMIT; }
    return ICAP_OK;
}

// cfg: -1 = choose by shape, 0 = 128 x 128 tiles (one workgroup per CU, 144 KB ring), 1 = 64 x 64 (two per CU)
int linear_x3dl_f32(const unsigned short* x3, const unsigned short* w3, const float* scale, const float* bias, const float* residual,
                    float* y, unsigned short* y3, int M, int K, int N, int act, int cfg, hipStream_t st) {
    if (!y && !y3) { set_error("linear_x3dl: no result buffer"); return ICAP_ERR_ARG; }
    const int rc = xd_args_ok("linear_x3dl", x3, w3, scale, bias, residual, y, y3, M, K, N);
    if (rc != ICAP_OK) return rc;
    X3dlParams p{x3, w3, scale, bias, residual, y, y3, M, N, K, act, 0, 0, nullptr, nullptr};
    if (cfg < 0) cfg = (long long)((M + 127) / 128) * ((N + 127) / 128) >= 256 ? 0 : 1;
    switch (cfg) {
        case 0: return xd_launch<128, 128, false>(p, st);
        case 1: return xd_launch<64, 64, false>(p, st);
        default: set_error("linear_x3dl: unknown tile configuration %d", cfg); return ICAP_ERR_ARG;
    }
}

// vocabulary projection with arg-max partials per 128-column tile ((N + 127) / 128 per row, as linear_f32x3_argmax_f32)
int linear_x3dl_argmax_f32(const unsigned short* x3, const unsigned short* w3, const float* scale, const float* bias, float* y,
                           float* part_val, int* part_idx, int M, int K, int N, hipStream_t st) {
    if (!part_val || !part_idx) { set_error("linear_x3dl_argmax: null partial buffers"); return ICAP_ERR_ARG; }
    const int rc = xd_args_ok("linear_x3dl_argmax", x3, w3, scale, bias, nullptr, y, nullptr, M, K, N);
    if (rc != ICAP_OK) return rc;
    X3dlParams p{x3, w3, scale, bias, nullptr, y, nullptr, M, N, K, ACT_NONE, 0, 0, part_val, part_idx};
    return M > 64 ? xd_launch<128, 128, true>(p, st) : xd_launch<64, 128, true>(p, st);
}

}  // namespace icap
