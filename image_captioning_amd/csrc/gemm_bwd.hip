// GEMMs of the backward pass, fp32 on v_mfma_f32_32x32x2_f32.
//
// The forward kernel (conv_igemm.hip) multiplies two operands whose rows are contiguous
// along the contraction index ("NT": Y = X W^T).  Back-propagation needs the other two
// forms, and transposing activations first would cost an extra pass over HBM:
//   NN  dX[M][N] = dY[M][K] . W[K][N]        A rows contiguous along k, B rows along n
//   TN  dW[M][N] = dY[K][M]^T . X[K][N]      both operands stored k-major; the contraction
//                                            runs over rows (batch x positions), often 10^5+
// An operand stored k-major is staged in LDS as [k][m] with 16-byte loads along m and
// read back one float per lane (lane = output row / column: consecutive addresses, no
// bank conflicts); an operand stored k-contiguous is staged as padded rows [m][k] and
// read back 16 bytes per lane.
//
// TN problems have few output tiles and a very long contraction, so the contraction is
// split over gridDim.z workgroups; each writes its partial tile to a workspace and a
// second kernel adds the partials in a fixed order (deterministic, no atomics).
#include "common.hpp"

namespace icap {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmParams {
    const float* a;
    const float* b;
    float* c;          // [M][ldc] result, or the partial workspace [splits][M][N] when splits > 1
    int M, N, K;
    int lda, ldb, ldc;
    int splits, k_per_split;
};

constexpr int GB_BM = 64, GB_BN = 64, GB_BK = 16;
constexpr int GB_LDK = GB_BK + 4;   // padded row of a k-contiguous operand tile
constexpr int GB_LDM = GB_BM + 4;   // padded row of a k-major operand tile (keeps 16-byte alignment)

// A_KM: A is stored [K][M] (k-major) instead of [M][K].  B is always [K][N].
template <bool A_KM>
__global__ __launch_bounds__(256) void gemm_bwd_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) float As[2][A_KM ? GB_BK * GB_LDM : GB_BM * GB_LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][GB_BK * GB_LDM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * GB_BM, n0 = blockIdx.x * GB_BN;
    const int k_lo = blockIdx.z * p.k_per_split;
    const int k_hi = min(p.K, k_lo + p.k_per_split);
    const int nk = (k_hi - k_lo + GB_BK - 1) / GB_BK;

    // staging coordinates: one float4 per thread and operand per K step
    //   k-major tile [16 k][64 m]: thread -> (k = tid / 16, m4 = tid % 16)
    //   k-contiguous tile [64 m][16 k]: thread -> (m = tid / 4, k4 = tid % 4)
    const int km_k = tid >> 4, km_c = (tid & 15) << 2;
    const int kc_m = tid >> 2, kc_k = (tid & 3) << 2;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);

    // 16-byte load with element-wise bounds along the contiguous index; rows and columns beyond the
    // matrix read a valid address and are zeroed
    auto load4 = [&](const float* base, int ld, int row, int row_lim, int col, int col_lim) -> float4 {
        const bool row_ok = row < row_lim;
        const float* src = base + (size_t)(row_ok ? row : 0) * ld;
        if (col + 3 < col_lim && (ld & 3) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(src + col);
            return row_ok ? v : z;
        }
        float4 v = z;
        if (row_ok) {
            if (col < col_lim) v.x = src[col];
            if (col + 1 < col_lim) v.y = src[col + 1];
            if (col + 2 < col_lim) v.z = src[col + 2];
            if (col + 3 < col_lim) v.w = src[col + 3];
        }
        return v;
    };
    float4 ra, rb;
    auto gload = [&](int t) {
        const int kb = k_lo + t * GB_BK;
        if (A_KM) ra = load4(p.a, p.lda, kb + km_k, k_hi, m0 + km_c, p.M);
        else ra = load4(p.a, p.lda, m0 + kc_m, p.M, kb + kc_k, k_hi);
        rb = load4(p.b, p.ldb, kb + km_k, k_hi, n0 + km_c, p.N);
    };
    auto lds_store = [&](int buf) {
        if (A_KM) *reinterpret_cast<float4*>(&As[buf][km_k * GB_LDM + km_c]) = ra;
        else *reinterpret_cast<float4*>(&As[buf][kc_m * GB_LDK + kc_k]) = ra;
        *reinterpret_cast<float4*>(&Bs[buf][km_k * GB_LDM + km_c]) = rb;
    };

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    if (nk > 0) {
        gload(0);
        lds_store(0);
    }
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < nk; ++t) {
        const bool more = t + 1 < nk;
        if (more) gload(t + 1);
        float af[8], bf[8];
        if (A_KM) {
#pragma unroll
            for (int s = 0; s < 8; ++s) af[s] = As[cur][(2 * s + lh) * GB_LDM + wm * 32 + lr];
        } else {
            // lane (lr, lh) takes k = 8*lh + s for step s; B below follows the same assignment
            const float* src = &As[cur][(wm * 32 + lr) * GB_LDK + 8 * lh];
            const float4 v0 = *reinterpret_cast<const float4*>(src);
            const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
            af[0] = v0.x; af[1] = v0.y; af[2] = v0.z; af[3] = v0.w;
            af[4] = v1.x; af[5] = v1.y; af[6] = v1.z; af[7] = v1.w;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = A_KM ? (2 * s + lh) : (8 * lh + s);
            bf[s] = Bs[cur][k * GB_LDM + wn * 32 + lr];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
        if (more) lds_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // a lane holds column n = lr and rows (e&3) + 8*(e>>2) + 4*lh of the wave's 32x32 block
    float* c = p.c + (p.splits > 1 ? (size_t)blockIdx.z * p.M * p.N : 0);
    const int ldc = p.splits > 1 ? p.N : p.ldc;
    const int n = n0 + wn * 32 + lr;
    if (n < p.N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            if (m < p.M) c[(size_t)m * ldc + n] = acc[e];
        }
    }
}

// TN form with wave tiles of several MFMA blocks: the weight gradients whose result is tens to hundreds of rows and
// columns (dW of the 1x1 convolutions from 56 x 56 down to 7 x 7, of the vocabulary projection and the LSTM gates).  Both
// operands are k-major, staged [BK k][BM m] with 16-byte loads and stores and read back one float per lane; per K pair
// a wave issues RM + RN LDS reads for RM x RN MFMAs (the 64 x 64 kernel above: 2 reads per MFMA), and two or three
// workgroups share a CU.  The four waves are laid out WM x (4 / WM) over the tile, so that a result as narrow as 24 x 144
// or 240 x 40 still fills whole 32 x 32 blocks along its long side.
// v or zeros, by value: `ok ? a[i] : z` between two named objects becomes a select of ADDRESSES, which pins both to memory
__device__ __forceinline__ float4 tn_keep(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

template <int BM, int BN, int WM, int BK>
__global__ __launch_bounds__(256, 2) void gemm_tn_tile_kernel(const GemmParams p) {
    constexpr int LDA = BM + 4, LDB = BN + 4, WN = 4 / WM;
    constexpr int RM = BM / (32 * WM), RN = BN / (32 * WN);          // 32 x 32 blocks per wave along m and n
    constexpr int PA = BK * BM / 1024, PB = BK * BN / 1024;          // float4 per thread and K step
    static_assert(RM >= 1 && RN >= 1 && PA >= 1 && PB >= 1 && BK % 2 == 0, "tile too small for 256 threads");
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int k_lo = blockIdx.z * p.k_per_split;
    const int k_hi = min(p.K, k_lo + p.k_per_split);
    const int nk = (k_hi - k_lo + BK - 1) / BK;
    // The host guarantees lda, ldb, M, N multiples of 4, so a 16-byte group is inside the matrix or outside it as a whole.
    // Every load is unconditional, from a clamped address, and zeroed afterwards: a branch around a load makes the
    // compiler wait for each load before the next.
    const float* a_src[PA];
    const float* b_src[PB];
    bool a_ok[PA], b_ok[PB];
    int a_k[PA], b_k[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int idx = tid + 256 * i, col = m0 + (idx % (BM / 4)) * 4;
        a_k[i] = idx / (BM / 4);
        a_ok[i] = col < p.M;
        a_src[i] = p.a + (a_ok[i] ? col : 0);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int idx = tid + 256 * i, col = n0 + (idx % (BN / 4)) * 4;
        b_k[i] = idx / (BN / 4);
        b_ok[i] = col < p.N;
        b_src[i] = p.b + (b_ok[i] ? col : 0);
    }
    // two register sets: the loads of step t + 2 are issued while step t is being multiplied, so a load has two steps of
    // matrix-core work (about 4 us) to arrive before it is written to LDS
    float4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
    auto gload = [&](int t, float4 (&ra)[PA], float4 (&rb)[PB]) {
        const int kb = k_lo + t * BK;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int row = kb + a_k[i];
            ra[i] = *reinterpret_cast<const float4*>(a_src[i] + (size_t)(row < k_hi ? row : k_lo) * p.lda);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int row = kb + b_k[i];
            rb[i] = *reinterpret_cast<const float4*>(b_src[i] + (size_t)(row < k_hi ? row : k_lo) * p.ldb);
        }
    };
    // the values of step t go to LDS, zeroed where they lie outside the matrix or this split's rows (the select sits here,
    // next to the store that has to wait for the load anyway, so the loads stay in flight during the previous step's MFMAs)
    auto lds_store = [&](int buf, int t, const float4 (&ra)[PA], const float4 (&rb)[PB]) {
        const int kb = k_lo + t * BK;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int idx = tid + 256 * i;
            const bool ok = a_ok[i] && kb + a_k[i] < k_hi;
            *reinterpret_cast<float4*>(&As[buf][(idx / (BM / 4)) * LDA + (idx % (BM / 4)) * 4]) = tn_keep(ok, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int idx = tid + 256 * i;
            const bool ok = b_ok[i] && kb + b_k[i] < k_hi;
            *reinterpret_cast<float4*>(&Bs[buf][(idx / (BN / 4)) * LDB + (idx % (BN / 4)) * 4]) = tn_keep(ok, rb[i]);
        }
    };

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    gload(0, ra0, rb0);
    gload(1, ra1, rb1);
    lds_store(0, 0, ra0, rb0);
    gload(2, ra0, rb0);
    __syncthreads();
    int cur = 0;
    // one step: multiply LDS buffer `cur` (step t), then write step t + 1 (in flight since step t - 1) to the other buffer
    // and issue the loads of step t + 3 into the registers it frees
    auto step = [&](int t, float4 (&ra)[PA], float4 (&rb)[PB]) {
        const float* as = &As[cur][lh * LDA + wm * (BM / WM) + lr];
        const float* bs = &Bs[cur][lh * LDB + wn * (BN / WN) + lr];
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            float af[RM], bf[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i) af[i] = as[2 * s * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < RN; ++j) bf[j] = bs[2 * s * LDB + 32 * j];
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        lds_store(cur ^ 1, t + 1, ra, rb);
        __builtin_amdgcn_sched_barrier(0);      // keep the new loads behind the stores: hoisted, they land in a third register set
                                                // and the copy back waits for them on the spot
        gload(t + 3, ra, rb);       // unconditional: past the last step it re-reads row k_lo (valid memory) and the values are dropped
        __syncthreads();
        cur ^= 1;
    };
    // steps run in pairs (one per register set); with an odd count the last one multiplies a buffer of zeros
    for (int t = 0; t < nk; t += 2) {
        step(t, ra1, rb1);
        step(t + 1, ra0, rb0);
    }

    float* c = p.c + (p.splits > 1 ? (size_t)blockIdx.z * p.M * p.N : 0);
    const int ldc = p.splits > 1 ? p.N : p.ldc;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int n = n0 + wn * (BN / WN) + 32 * j + lr;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / WM) + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m < p.M) c[(size_t)m * ldc + n] = acc[i][j][e];
            }
    }
}

struct TnTile {
    int bm, bn;
    void (*fn)(GemmParams);
};
// in order of preference when two tiles pad the result to the same area
static const TnTile kTnTiles[] = {
    {128, 128, gemm_tn_tile_kernel<128, 128, 2, 32>}, {256, 64, gemm_tn_tile_kernel<256, 64, 2, 16>},
    {64, 256, gemm_tn_tile_kernel<64, 256, 2, 16>},   {128, 64, gemm_tn_tile_kernel<128, 64, 2, 32>},
    {256, 32, gemm_tn_tile_kernel<256, 32, 4, 32>},   {32, 256, gemm_tn_tile_kernel<32, 256, 1, 32>},
};

// which tile the TN form uses for an M x N result: an index into kTnTiles (least padded area), or -1 for the 64 x 64
// kernel (unaligned operands, results too small to fill a third of any tile)
static int tn_tile(int M, int N, int lda = 0, int ldb = 0) {
    if ((M | N | lda | ldb) & 3) return -1;
    int best = -1;
    long long best_area = 0;
    for (int i = 0; i < (int)(sizeof(kTnTiles) / sizeof(kTnTiles[0])); ++i) {
        const TnTile& t = kTnTiles[i];
        const long long area = (long long)((M + t.bm - 1) / t.bm) * t.bm * ((N + t.bn - 1) / t.bn) * t.bn;
        if (best < 0 || area < best_area) { best = i; best_area = area; }
    }
    return 3 * (long long)M * N >= best_area ? best : -1;
}

static int pick_splits_tile(int M, int N, int K, int bm, int bn) {
    const long long tiles = (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    if (tiles >= 512 || K <= 256) return 1;
    long long s = 512 / tiles > 0 ? 512 / tiles : 1;     // one round of two resident workgroups per CU: a second, partial round costs a whole pass
    const long long max_s = K / 128 > 0 ? K / 128 : 1;   // at least 8 contraction steps per split
    if (s > max_s) s = max_s;
    if (s > 768) s = 768;
    return (int)(s < 1 ? 1 : s);
}

// c[m][n] = (accumulate ? c[m][n] : 0) + sum_s part[s][m][n], fixed order; sixteen independent loads in flight per lane
// (a plain loop chains one memory latency per split)
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int splits, int M, int N,
                                                                 float* __restrict__ c, int ldc, int accumulate) {
    const long long total = (long long)M * N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i - (long long)m * N);
        float s = accumulate ? c[(size_t)m * ldc + n] : 0.f;
        for (int k0 = 0; k0 < splits; k0 += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(k0 + u < splits ? k0 + u : k0) * total + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += k0 + u < splits ? v[u] : 0.f;
        }
        c[(size_t)m * ldc + n] = s;
    }
}

static int pick_splits(int M, int N, int K) {
    const long long tiles = (long long)((M + GB_BM - 1) / GB_BM) * ((N + GB_BN - 1) / GB_BN);
    if (tiles >= 512 || K <= 256) return 1;
    long long s = (1024 + tiles - 1) / tiles;
    const long long max_s = K / 128 > 0 ? K / 128 : 1;   // at least 128 contraction steps per split
    if (s > max_s) s = max_s;
    if (s > 256) s = 256;
    return (int)(s < 1 ? 1 : s);
}

// ---------------------------------------------------------------- tall-and-skinny TN form
// C[M][N] = A^T B with A [K][M], B [K][N], K in the hundreds of thousands to millions and M, N at most a few 32-wide
// blocks: the weight gradient of the thin 1x1 convolutions at high resolution.  The problem is a streaming reduction
// over K (both operands are read once, the result is a few KB), so nothing is staged: a wave owns a contiguous range of
// K rows and the whole MB x NB block tile in MFMA accumulators, and loads its operands straight in the MFMA's layout -
// lane (r, h) reads A[k + h][m0 + r] (32 lanes = one 128-byte row segment) - U row pairs in flight at a time.  The four
// waves of a workgroup add their tiles in LDS in wave order, workgroups write partial tiles [chunk][M][N] and
// gemm_tn_finish_kernel adds the chunks in a fixed order.
typedef float f32x16s __attribute__((ext_vector_type(16)));
constexpr int SK_U = 8;          // row pairs in flight per wave
constexpr int SK_CHUNKS = 512;   // workgroups along K

template <int MB, int NB>
__global__ __launch_bounds__(256) void gemm_tn_skinny_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                             int ldb, int K, int M, int N, int rows_per_wave,
                                                             float* __restrict__ part) {
    __shared__ float tile[MB * 32][NB * 32 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int chunk = blockIdx.x;
    const int mt = blockIdx.y * MB * 32, nt = blockIdx.z * NB * 32;
    const int r_lo = (chunk * 4 + wave) * rows_per_wave;
    const int r_hi = min(K, r_lo + rows_per_wave);
    f32x16s acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    bool a_ok[MB], b_ok[NB];
    int a_col[MB], b_col[NB];
#pragma unroll
    for (int i = 0; i < MB; ++i) { a_ok[i] = mt + i * 32 + lr < M; a_col[i] = a_ok[i] ? mt + i * 32 + lr : 0; }
#pragma unroll
    for (int j = 0; j < NB; ++j) { b_ok[j] = nt + j * 32 + lr < N; b_col[j] = b_ok[j] ? nt + j * 32 + lr : 0; }
    for (int r = r_lo; r < r_hi; r += 2 * SK_U) {
        float av[SK_U][MB], bv[SK_U][NB];
#pragma unroll
        for (int u = 0; u < SK_U; ++u) {
            const int row = r + 2 * u + lh;
            const bool ok = row < r_hi;
            const size_t rc = (size_t)(ok ? row : r_lo);          // always a valid row
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                av[u][i] = a[rc * lda + a_col[i]];
                if (!(ok && a_ok[i])) av[u][i] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                bv[u][j] = b[rc * ldb + b_col[j]];
                if (!(ok && b_ok[j])) bv[u][j] = 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < SK_U; ++u)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
    }
    // the four waves meet in LDS in wave order (a lane holds column lr and rows (e&3) + 8*(e>>2) + 4*lh of a block)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float& t = tile[i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh][j * 32 + lr];
                        t = (w == 0 ? 0.f : t) + acc[i][j][e];
                    }
        }
        __syncthreads();
    }
    float* dst = part + (size_t)chunk * M * N;
    for (int idx = threadIdx.x; idx < MB * 32 * NB * 32; idx += 256) {
        const int mi = idx / (NB * 32), ni = idx - mi * (NB * 32);
        if (mt + mi < M && nt + ni < N) dst[(size_t)(mt + mi) * N + nt + ni] = tile[mi][ni];
    }
}

// c[m][n] (+)= sum over chunks of part[chunk][m][n]: 8 outputs per workgroup, 32 lanes per output each adding every 32nd
// chunk with sixteen loads in flight; fixed order.
__global__ __launch_bounds__(256) void gemm_tn_finish_kernel(const float* __restrict__ part, int nchunks, int M, int N,
                                                             float* __restrict__ c, int ldc, int accumulate) {
    __shared__ float red[32][8];
    const int cx = threadIdx.x & 7, ky = threadIdx.x >> 3;
    const int total = M * N;
    const int i = blockIdx.x * 8 + cx;
    float s = 0.f;
    if (i < total) {
        for (int k0 = ky; k0 < nchunks; k0 += 32 * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = k0 + 32 * u;
                v[u] = k < nchunks ? part[(size_t)k * total + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
    }
    red[ky][cx] = s;
    __syncthreads();
    if (ky == 0 && i < total) {
        float t = red[0][cx];
#pragma unroll
        for (int k = 1; k < 32; ++k) t += red[k][cx];
        float* o = c + (size_t)(i / N) * ldc + i % N;
        *o = (accumulate ? *o : 0.f) + t;
    }
}

// The skinny form pays when the contraction is long and the result is at most a few blocks.
static bool skinny_tn(int M, int N, int K) {
    return K >= 32768 && (long long)((M + 31) / 32) * ((N + 31) / 32) <= 16 && (long long)M * N <= (1 << 20) &&
           ((long long)((M + 31) / 32) * ((N + 31) / 32) <= 3 || tn_tile(M, N) < 0);
}

// A wave keeps 2 SK_U row pairs of loads in flight and rows are short (64-640 bytes), so bytes in flight come from
// resident waves: 2048 workgroups (8 per CU) for tiles of one or two blocks, 1024 for the wider ones (registers).
static int skinny_chunks(int K, int blocks) {
    int chunks = blocks <= 2 ? 4 * SK_CHUNKS : 2 * SK_CHUNKS;
    while (chunks > 1 && K / (chunks * 4) < 4 * SK_U) chunks >>= 1;    // at least two batches of rows per wave
    return chunks;
}

template <int MB, int NB>
static void skinny_launch(const float* a, int lda, const float* b, int ldb, int K, int M, int N, int chunks, float* ws, hipStream_t st) {
    int rpw = (K + chunks * 4 - 1) / (chunks * 4);
    rpw = (rpw + 1) & ~1;                                   // row pairs must not straddle two waves
    const dim3 grid(chunks, (M + MB * 32 - 1) / (MB * 32), (N + NB * 32 - 1) / (NB * 32));
    hipLaunchKernelGGL((gemm_tn_skinny_kernel<MB, NB>), grid, dim3(256), 0, st, a, lda, b, ldb, K, M, N, rpw, ws);
}

// ---------------------------------------------------------------- weight gradient of a few-channel convolution (the stem)
// dW[co][(r*KW + q)*Cv + c] = sum_m dz[m][co] * x[b, ho*S - pad_t + r, wo*S - pad_l + q, c],  m = (b*Ho + ho)*Wo + wo, c < Cv <= C
// (C: stored channels per pixel, Cv: the real ones).  It is the skinny TN form with the second operand GATHERED from the
// image: no patch matrix is written and read back (for the 3-channel 3x3 stride-2 stem at B = 256 that matrix is 460 MB
// against 205 MB of image).  Cout <= 32 and KH*KW*Cv <= 32: one MFMA block per wave.  A lane's column fixes its tap and
// channel; its row's (b, ho, wo) comes from one division per batch of rows and is stepped from there.
__global__ __launch_bounds__(256) void conv_wgrad_gather_kernel(const float* __restrict__ dz, int ldz, const float* __restrict__ x,
                                                                int H, int W, int C, int Cv, int KW, int S, int pad_t, int pad_l,
                                                                int Ho, int Wo, int K, int M, int N, int rows_per_wave,
                                                                float* __restrict__ part) {
    __shared__ float tile[32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int chunk = blockIdx.x;
    const int r_lo = (chunk * 4 + wave) * rows_per_wave;
    const int r_hi = min(K, r_lo + rows_per_wave);
    f32x16s acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const bool a_ok = lr < M, b_ok = lr < N;
    const int a_col = a_ok ? lr : 0;
    const int n = b_ok ? lr : 0, tap = n / Cv, c = n - tap * Cv, kr = tap / KW, kq = tap - kr * KW;
    const int dr = kr - pad_t, dq = kq - pad_l;
    for (int r = r_lo; r < r_hi; r += 2 * SK_U) {
        const int row0 = r + lh;
        int wo = row0 % Wo;
        const int t = row0 / Wo;
        int ho = t % Ho, b = t / Ho;
        float av[SK_U], bv[SK_U];
#pragma unroll
        for (int u = 0; u < SK_U; ++u) {
            const int row = r + 2 * u + lh;
            const bool ok = row < r_hi;
            const int hi = ho * S + dr, wi = wo * S + dq;
            const bool inb = ok && b_ok && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
            const size_t xo = inb ? (((size_t)b * H + hi) * W + wi) * C + c : 0;      // always a valid address
            av[u] = dz[(size_t)(ok ? row : r_lo) * ldz + a_col];
            bv[u] = x[xo];
            if (!(ok && a_ok)) av[u] = 0.f;
            if (!inb) bv[u] = 0.f;
            wo += 2;
            while (wo >= Wo) {
                wo -= Wo;
                if (++ho >= Ho) { ho = 0; ++b; }
            }
        }
#pragma unroll
        for (int u = 0; u < SK_U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    // the four waves meet in LDS in wave order (a lane holds column lr and rows (e&3) + 8*(e>>2) + 4*lh of the block)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float& tv = tile[(e & 3) + 8 * (e >> 2) + 4 * lh][lr];
                tv = (w == 0 ? 0.f : tv) + acc[e];
            }
        }
        __syncthreads();
    }
    float* dst = part + (size_t)chunk * M * N;
    for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
        const int mi = idx >> 5, ni = idx & 31;
        if (mi < M && ni < N) dst[(size_t)mi * N + ni] = tile[mi][ni];
    }
}

long long conv_wgrad_gather_ws_floats(int B, int Ho, int Wo, int M, int N) {
    const long long K = (long long)B * Ho * Wo;
    if (K <= 0 || K >= (1LL << 31) || M <= 0 || N <= 0) return 0;
    return (long long)skinny_chunks((int)K, 1) * M * N;
}

int conv_wgrad_gather_f32(const float* dz, int ldz, const float* x, int B, int H, int W, int C, int Cv, int KH, int KW, int S,
                          int pad_t, int pad_l, int Ho, int Wo, int Cout, float* dw, float* ws, hipStream_t st) {
    if (!dz || !x || !dw || !ws || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Cv <= 0 || Cv > C || KH <= 0 || KW <= 0 || S <= 0 || Ho <= 0 ||
        Wo <= 0 || Cout <= 0 || ldz < Cout || pad_t < 0 || pad_l < 0) {
        set_error("conv_wgrad_gather: bad arguments");
        return ICAP_ERR_ARG;
    }
    const int N = KH * KW * Cv;
    const long long K = (long long)B * Ho * Wo;
    if (Cout > 32 || N > 32 || K >= (1LL << 31)) {
        set_error("conv_wgrad_gather: Cout %d and KH*KW*Cv %d must be <= 32 (patch matrix + icap_gemm_bwd_f32 otherwise)", Cout, N);
        return ICAP_ERR_LIMIT;
    }
    // the last output position must read inside the padded extent the forward pass used
    if ((long long)(Ho - 1) * S - pad_t >= H || (long long)(Wo - 1) * S - pad_l >= W) { set_error("conv_wgrad_gather: output larger than the input allows"); return ICAP_ERR_ARG; }
    const int chunks = skinny_chunks((int)K, 1);
    int rpw = (int)((K + chunks * 4 - 1) / (chunks * 4));
    rpw = (rpw + 1) & ~1;                                   // row pairs must not straddle two waves
    hipLaunchKernelGGL(conv_wgrad_gather_kernel, dim3(chunks), dim3(256), 0, st, dz, ldz, x, H, W, C, Cv, KW, S, pad_t, pad_l, Ho, Wo, (int)K,
                       Cout, N, rpw, ws);
    hipLaunchKernelGGL(gemm_tn_finish_kernel, dim3((Cout * N + 7) / 8), dim3(256), 0, st, ws, chunks, Cout, N, dw, N, 0);
    return launch_status("conv_wgrad_gather");
}

// ---------------------------------------------------------------- short contraction (the reference's loader batch of 12)
// TN, K <= 16:  C[M][N] (+)= A[K][M]^T . B[K][N] - a sum of K outer products; a thread owns one row and four columns
template <int KR>
__global__ __launch_bounds__(256) void gemm_tn_fewk_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                           float* __restrict__ c, int ldc, int M, int N, int K, int accumulate) {
    const int nq4 = (N + 3) >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)M * nq4) return;
    const int m = (int)(idx / nq4), n = (int)(idx - (long long)m * nq4) << 2;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        if (k < K) {
            const float av = a[(size_t)k * lda + m];
            const float4 bv = *reinterpret_cast<const float4*>(b + (size_t)k * ldb + n);
            acc.x = fmaf(av, bv.x, acc.x); acc.y = fmaf(av, bv.y, acc.y); acc.z = fmaf(av, bv.z, acc.z); acc.w = fmaf(av, bv.w, acc.w);
        }
    }
    float4* o = reinterpret_cast<float4*>(c + (size_t)m * ldc + n);
    if (accumulate) { const float4 p = *o; acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w; }
    *o = acc;
}

static bool fewrow_ok(const float* b, const float* c, int ldb, int ldc, int N) {
    return (N & 3) == 0 && (ldb & 3) == 0 && (ldc & 3) == 0 && aligned16(b) && aligned16(c);
}

// floats of workspace gemm_bwd_f32 needs for this problem (0 when the contraction is not split and the result
// is not accumulated onto C); sized for either operand layout
long long gemm_bwd_ws_floats(int M, int N, int K, int accumulate) {
    int s = pick_splits(M, N, K);
    const int tile = tn_tile(M, N);
    if (tile >= 0) {
        const int st = pick_splits_tile(M, N, K, kTnTiles[tile].bm, kTnTiles[tile].bn);
        if (st > s) s = st;
    }
    long long need = (s > 1 || accumulate) ? (long long)s * M * N : 0;
    if (skinny_tn(M, N, K)) {
        const long long sk = (long long)skinny_chunks(K, 1) * M * N;
        if (sk > need) need = sk;
    }
    return need;
}

// C[M][N] (+)= op(A) . B with B stored [K][N]; op(A) = A stored [M][K] (a_kmajor = 0) or A stored [K][M] (1).
int gemm_bwd_f32(const float* a, int lda, int a_kmajor, const float* b, int ldb, float* c, int ldc, int M, int N, int K,
                 int accumulate, float* ws, hipStream_t st) {
    if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0 || ldb < N || ldc < N || (a_kmajor ? lda < M : lda < K)) {
        set_error("gemm_bwd: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (!aligned16(a) || !aligned16(b)) { set_error("gemm_bwd: operands must be 16-byte aligned"); return ICAP_ERR_ALIGN; }
    if (a_kmajor && K <= 16 && fewrow_ok(b, c, ldb, ldc, N)) {
        const long long threads = (long long)M * (N / 4);
        hipLaunchKernelGGL(gemm_tn_fewk_kernel<16>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a, lda, b, ldb, c, ldc, M, N,
                           K, accumulate);
        return launch_status("gemm_bwd (short contraction)");
    }
    if (a_kmajor && skinny_tn(M, N, K)) {
        if (!ws) { set_error("gemm_bwd: workspace required (gemm_bwd_ws_floats)"); return ICAP_ERR_ARG; }
        const int mb = (M + 31) / 32, nb = (N + 31) / 32;
        const int wide = mb >= nb ? mb : nb;
        const int chunks = skinny_chunks(K, wide > 5 ? ((wide + 1) / 2 > 5 ? 4 : (wide + 1) / 2) : wide);
        // one of the two tile extents is a single block; the other covers up to five, the grid tiles the rest
        if (mb >= nb) {
            const int t = mb > 5 ? (mb + 1) / 2 > 5 ? 4 : (mb + 1) / 2 : mb;
            switch (t) {
                case 1: skinny_launch<1, 1>(a, lda, b, ldb, K, M, N, chunks, ws, st); break;
                case 2: skinny_launch<2, 1>(a, lda, b, ldb, K, M, N, chunks, ws, st); break;
                case 3: skinny_launch<3, 1>(a, lda, b, ldb, K, M, N, chunks, ws, st); break;
                case 4: skinny_launch<4, 1>(a, lda, b, ldb, K, M, N, chunks, ws, st); break;
                default: skinny_launch<5, 1>(a, lda, b, ldb, K, M, N, chunks, ws, st); break;
            }
        } else {
            const int t = nb > 5 ? (nb + 1) / 2 > 5 ? 4 : (nb + 1) / 2 : nb;
            switch (t) {
                case 2: skinny_launch<1, 2>(a, lda, b, ldb, K, M, N, chunks, ws, st); break;
                case 3: skinny_launch<1, 3>(a, lda, b, ldb, K, M, N, chunks, ws, st); break;
                case 4: skinny_launch<1, 4>(a, lda, b, ldb, K, M, N, chunks, ws, st); break;
                default: skinny_launch<1, 5>(a, lda, b, ldb, K, M, N, chunks, ws, st); break;
            }
        }
        hipLaunchKernelGGL(gemm_tn_finish_kernel, dim3((M * N + 7) / 8), dim3(256), 0, st, ws, chunks, M, N, c, ldc, accumulate);
        return launch_status("gemm_bwd (skinny)");
    }
    const int tile = a_kmajor ? tn_tile(M, N, lda, ldb) : -1;
    const int bm = tile >= 0 ? kTnTiles[tile].bm : GB_BM, bn = tile >= 0 ? kTnTiles[tile].bn : GB_BN;
    int splits = tile >= 0 ? pick_splits_tile(M, N, K, bm, bn) : pick_splits(M, N, K);
    if (splits > 1 && !ws) { set_error("gemm_bwd: workspace required (gemm_bwd_ws_floats)"); return ICAP_ERR_ARG; }
    GemmParams p{a, b, c, M, N, K, lda, ldb, ldc, splits, 0};
    int kps = (K + splits - 1) / splits;
    kps = (kps + GB_BK - 1) / GB_BK * GB_BK;
    splits = (K + kps - 1) / kps;
    p.splits = splits;
    p.k_per_split = kps;
    const bool direct = splits == 1 && !accumulate;
    if (!direct) {
        if (!ws && splits == 1) { set_error("gemm_bwd: accumulation needs a workspace of M*N floats"); return ICAP_ERR_ARG; }
        p.c = ws;
        if (splits == 1) { p.splits = 2; }   // route the single partial through the workspace layout [1][M][N]
    }
    const dim3 grid((N + bn - 1) / bn, (M + bm - 1) / bm, splits);
    if (tile >= 0) hipLaunchKernelGGL(kTnTiles[tile].fn, grid, dim3(256), 0, st, p);
    else if (a_kmajor) hipLaunchKernelGGL(gemm_bwd_kernel<true>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(gemm_bwd_kernel<false>, grid, dim3(256), 0, st, p);
    if (!direct) {
        const long long total = (long long)M * N;
        long long blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, ws, splits, M, N, c, ldc, accumulate);
    }
    return launch_status("gemm_bwd");
}

}  // namespace icap
