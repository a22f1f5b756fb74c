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
// Tiling: BM x BN outputs per 256-thread workgroup (4 waves).  The waves are
// arranged WM x WN over the tile and KSPLIT over the K slice (WM*WN*KSPLIT = 4):
// big problems use 2x2 or 4x1 waves on one 16-wide K slice, problems with only a
// few hundred rows (decode-time linears) use a 32x32 tile with the four waves
// splitting a 64-wide K slice and a fixed-order reduction through LDS, so that
// even [256 x 512] outputs put >100 workgroups on the chip.
// Operand tiles are staged global -> registers -> LDS (double buffered, one
// barrier per K step); LDS rows are padded by 16 B so ds_read_b128 fragment
// reads are conflict free.  The epilogue transposes the accumulators through LDS
// and writes 16 B per lane along the channel axis (full 128-B lines per row).
// Workgroup ids are remapped so that tiles sharing activation rows land on the
// same XCD (its L2 then serves the re-reads).
#include <map>
#include <mutex>
#include <tuple>
#include <stdlib.h>
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
    int tiles_n, tiles_m, m_fast, nblk, vec_ok;
    float* amax_val;        // optional [M][tiles_n]: per-tile row maxima of the outputs (arg-max partials)
    int* amax_idx;          // their column indices
    float* stat_part;       // optional [P][2][Cout]: column sum and squared deviations of every BM-row tile of the result
};

// Phase time stamps for the tuning probe under tools/ (compiled in only there): wave 0 of the first 64 workgroups,
// K steps 0..31 of their first tile, four stamps per step.
#ifdef ICAP_IGEMM_PROF
__device__ unsigned long long g_igemm_prof[64 * 32 * 4];
#define IG_STAMP(t, k)                                                                                       \
    do {                                                                                                     \
        if (threadIdx.x == 0 && blockIdx.x < 64 && (t) < 32 && tl == (int)blockIdx.x)                        \
            g_igemm_prof[(blockIdx.x * 32 + (t)) * 4 + (k)] = __builtin_amdgcn_s_memtime();                  \
    } while (0)
#else
#define IG_STAMP(t, k)
#endif

constexpr int BK = 16;       // K elements one wave consumes per step
constexpr int EP_ROWS = 64;  // output rows per epilogue pass

template <int S>
struct SetTag { static constexpr int value = S; };

template <int S, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (S < N) {
        f(SetTag<S>{});
        static_for<S + 1, N>(f);
    }
}

// PF: K slices in flight in registers (the number of K steps must be a multiple of it).  A
// workgroup that is alone on its CU - few-tile problems such as the decoder GEMMs at 256 rows -
// otherwise stalls every step for the part of the memory latency one step of MFMAs does not cover.
// STATS: the epilogue also leaves the column statistics of the result (its own instantiation: the moments it carries cost
// registers - and with them a resident wave per SIMD on the thin-result tiles - that the other launches should not pay).
template <int BM, int BN, int WM, int WN, int KSPLIT, int KU, bool ONE_BY_ONE, bool GATE, bool VEC, int PF = 1, bool AMAX = false, bool FRAG = false,
          bool STATS = false>
__global__ __launch_bounds__(64 * WM * WN * KSPLIT) void conv_igemm_kernel(const ConvParams p) {
    constexpr int NT = 64 * WM * WN * KSPLIT;   // threads: one wave per (row block, column block, K split)
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int RM = TM / 32, RN = TN / 32;
    constexpr int CPR = 4 * KSPLIT * KU;      // float4 chunks per staged row
    constexpr int BKS = BK * KSPLIT * KU;     // K elements staged per step (each wave consumes KU 16-wide slices of it)
    constexpr int LDK = BKS + 4;              // padded LDS row (floats)
    constexpr int RPI = NT / CPR;             // rows staged per pass of the workgroup
    constexpr int RA = (BM + RPI - 1) / RPI;
    constexpr int RB = (BN + RPI - 1) / RPI;
    constexpr int EP_LD = BN + 4;
    constexpr int EP_R = BM < EP_ROWS ? BM : EP_ROWS;
    constexpr int STAGE_FLOATS = 2 * (BM + BN) * LDK;
    constexpr int EP_FLOATS = KSPLIT * EP_R * EP_LD;
    constexpr int SMEM_FLOATS = STAGE_FLOATS > EP_FLOATS ? STAGE_FLOATS : EP_FLOATS;
    static_assert(NT >= 256 && NT <= 1024 && NT % CPR == 0, "4 to 16 waves per workgroup");
    static_assert(TM % 32 == 0 && TN % 32 == 0, "wave tile is built from 32x32 blocks");

    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float* As = smem;                       // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;        // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int ks = wave / (WM * WN);            // K-split index of this wave
    const int wm = (wave % (WM * WN)) / WN;
    const int wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int kcx = tid % CPR;   // which float4 of the staged K slice
    const int srow = tid / CPR;

    // ---- per-tile staging coordinates ----
    int m0 = 0, n0 = 0;
    const float* a_base[RA];
    int a_hi0[RA], a_wi0[RA], a_b[RA];
    bool a_ok[RA];
    const float* b_base[RB];
    bool b_ok[RB];

    // Workgroups are persistent: workgroup g handles tiles g, g + G, g + 2G, ... (G = gridDim.x, a
    // multiple of 8).  Workgroups g and g + 8 share an XCD, so a tile index t always runs on XCD
    // group t % 8; the remap below gives each group a contiguous range of tiles (n fastest), so
    // tiles that re-read the same activation rows share an L2.
    auto setup_tile = [&](int tl) {
        const int xcd = tl & 7, q = p.nblk >> 3, r = p.nblk & 7;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (tl >> 3);
        // tiles that share an XCD's L2 should re-read the LARGER operand: n fastest when the activations dominate
        // (convolutions), m fastest when the weights do (a vocabulary projection at a few hundred rows)
        const int tile_n = p.m_fast ? logical / p.tiles_m : logical % p.tiles_n;
        const int tile_m = p.m_fast ? logical % p.tiles_m : logical / p.tiles_n;
        m0 = tile_m * BM;
        n0 = tile_n * BN;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int row = srow + RPI * i;
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
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = srow + RPI * i;
            const int n = n0 + row;
            b_ok[i] = (row < BN) && (n < p.Cout);
            b_base[i] = p.w + (size_t)(b_ok[i] ? n : 0) * p.K;
        }
    };

    // Staging registers for the K slice in flight.  Every load below is unconditional (out-of-range
    // elements read a valid dummy address and are zeroed when the slice is written to LDS) and nothing
    // consumes a loaded value before lds_store(), so the loads of the next slice - or of the next
    // tile's first slice - stay in flight behind the MFMAs and the epilogue instead of being waited
    // for where they are issued.
    constexpr int NSET = FRAG ? 2 : PF;
    static_assert(!FRAG || PF == 1, "the fragment pipeline has its own two-slice prefetch");
    float4 ra[NSET][RA], rb[NSET][RB], rg[NSET][GATE ? RA : 1];
    unsigned okmask_s[NSET];   // bit i: ra[i] valid, bit 16+i: rb[i] valid

    auto gload = [&](int t, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        const int kk = t * BKS + kcx * 4;
        const bool kin = kk < p.K;
        int r = 0, q = 0, c = kk;
        if (!ONE_BY_ONE) {
            const int tap = kk / p.Cin;
            c = kk - tap * p.Cin;
            r = tap / p.KW;
            q = tap - r * p.KW;
        }
        unsigned okmask = 0;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int hi = a_hi0[i] + r, wi = a_wi0[i] + q;
            const bool ok = kin && a_ok[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const float* src = ok ? a_base[i] + ((size_t)hi * p.W + wi) * p.Cin + c : p.x;
            ra[S][i] = *reinterpret_cast<const float4*>(src);
            if (GATE) rg[S][i] = *reinterpret_cast<const float4*>(ok ? p.gate + (size_t)a_b[i] * p.Cin + c : p.gate);
            okmask |= ok ? (1u << i) : 0u;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const bool ok = kin && b_ok[i];
            rb[S][i] = *reinterpret_cast<const float4*>(ok ? b_base[i] + kk : p.w);
            okmask |= ok ? (1u << (16 + i)) : 0u;
        }
        okmask_s[S] = okmask;
    };
    auto lds_store = [&](int buf, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        const unsigned okmask = okmask_s[S];
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int row = srow + RPI * i;
            float4 v = ra[S][i];
            if (GATE) { v.x *= rg[S][i].x; v.y *= rg[S][i].y; v.z *= rg[S][i].z; v.w *= rg[S][i].w; }
            if (!((okmask >> i) & 1u)) v = z;
            if (BM % RPI == 0 || row < BM) *reinterpret_cast<float4*>(&As[(buf * BM + row) * LDK + kcx * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = srow + RPI * i;
            const float4 v = ((okmask >> (16 + i)) & 1u) ? rb[S][i] : z;
            if (BN % RPI == 0 || row < BN) *reinterpret_cast<float4*>(&Bs[(buf * BN + row) * LDK + kcx * 4]) = v;
        }
    };

    const int nk = (p.K + BKS - 1) / BKS;
    const int G = gridDim.x;
    int tl = blockIdx.x;
    if (tl >= p.nblk) return;
    setup_tile(tl);
    gload(0, SetTag<0>{});
    static_for<1, NSET>([&](auto st) { gload(decltype(st)::value, st); });

    float* ep = smem;  // [KSPLIT][EP_R][EP_LD]; aliases the staging buffers, used only between K loops
    constexpr int BN4 = BN / 4;
    constexpr int EP_RPI = NT / BN4;
    constexpr int NPASS = (BM + EP_ROWS - 1) / EP_ROWS;
    const int en4 = tid % BN4, erow = tid / BN4;

    while (true) {
        // registers hold K slices 0..NSET-1 of tile tl, staging coordinates describe tile tl
        if constexpr (!FRAG) {
            lds_store(0, SetTag<0>{});
            __syncthreads();
        }
        const int em0 = m0, en0 = n0;
        const int next = tl + G;
        const bool has_next = next < p.nblk;

        f32x16 acc[RM][RN];
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        int cur = 0;
        // MFMAs over the staged slice in LDS buffer `cur`.
        // Lane (lr, lh) owns k = 8*lh + s of a 16-wide slice for MFMA step s: any k -> (step, half)
        // assignment is valid as long as A and B agree.  Wave ks consumes slices ks*KU .. ks*KU+KU-1.
        auto compute = [&]() {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                float af[RM][8], bf[RN][8];
                const int koff = (ks * KU + u) * BK + 8 * lh;
#pragma unroll
                for (int i = 0; i < RM; ++i) {
                    const float* src = &As[(cur * BM + wm * TM + i * 32 + lr) * LDK + koff];
                    const float4 v0 = *reinterpret_cast<const float4*>(src);
                    const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
                    af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
                    af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
                }
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    const float* src = &Bs[(cur * BN + wn * TN + j * 32 + lr) * LDK + koff];
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
            }
        };

        if constexpr (FRAG) {
            // Fragment pipeline: the MFMA operands of step t+1 are read from LDS into a second register set while
            // the MFMAs of step t run from the first, so a step is [issue LDS reads] [MFMAs back to back] [move slice
            // t+2 from its staging registers to the LDS buffer slice t's fragments came from] [fetch slice t+4 into
            // those staging registers] [barrier]: no MFMA waits for an LDS read issued in its own step, and a fetched
            // slice has two steps to arrive.  Needs an even number (>= 4) of K steps; the steady-state loop is
            // branch-free, the last four steps are peeled and fetch the next tile's slices 0 and 1.
            float fa[2][KU][RM][8], fb[2][KU][RN][8];
            auto read_frags = [&](auto buf_tag, auto par_tag) {
                constexpr int BUF = decltype(buf_tag)::value, PAR = decltype(par_tag)::value;
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    const int koff = (ks * KU + u) * BK + 8 * lh;
#pragma unroll
                    for (int i = 0; i < RM; ++i) {
                        const float* src = &As[(BUF * BM + wm * TM + i * 32 + lr) * LDK + koff];
                        const float4 v0 = *reinterpret_cast<const float4*>(src);
                        const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
                        fa[PAR][u][i][0] = v0.x; fa[PAR][u][i][1] = v0.y; fa[PAR][u][i][2] = v0.z; fa[PAR][u][i][3] = v0.w;
                        fa[PAR][u][i][4] = v1.x; fa[PAR][u][i][5] = v1.y; fa[PAR][u][i][6] = v1.z; fa[PAR][u][i][7] = v1.w;
                    }
#pragma unroll
                    for (int j = 0; j < RN; ++j) {
                        const float* src = &Bs[(BUF * BN + wn * TN + j * 32 + lr) * LDK + koff];
                        const float4 v0 = *reinterpret_cast<const float4*>(src);
                        const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
                        fb[PAR][u][j][0] = v0.x; fb[PAR][u][j][1] = v0.y; fb[PAR][u][j][2] = v0.z; fb[PAR][u][j][3] = v0.w;
                        fb[PAR][u][j][4] = v1.x; fb[PAR][u][j][5] = v1.y; fb[PAR][u][j][6] = v1.z; fb[PAR][u][j][7] = v1.w;
                    }
                }
            };
            auto mma = [&](auto par_tag) {
                constexpr int PAR = decltype(par_tag)::value;
#pragma unroll
                for (int u = 0; u < KU; ++u)
#pragma unroll
                    for (int s = 0; s < 8; ++s)
#pragma unroll
                        for (int i = 0; i < RM; ++i)
#pragma unroll
                            for (int j = 0; j < RN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[PAR][u][i][s], fb[PAR][u][j][s], acc[i][j], 0, 0, 0);
            };
            using T0 = SetTag<0>;
            using T1 = SetTag<1>;
            // staging sets 0 / 1 hold slices 0 / 1 (fetched during the previous tile's last steps and epilogue)
            lds_store(0, T0{});
            gload(2, T0{});
            __syncthreads();
            read_frags(T0{}, T0{});
            lds_store(1, T1{});
            gload(3, T1{});
            __syncthreads();
            // step t (even): fragments(t) in set 0, LDS buffer 1 = slice t+1, staging 0 / 1 = slices t+2 / t+3
            for (int t = 0; t + 6 <= nk; t += 2) {
                IG_STAMP(t, 0);
                read_frags(T1{}, T1{});
                __builtin_amdgcn_sched_barrier(0);
                mma(T0{});
                __builtin_amdgcn_sched_barrier(0);
                IG_STAMP(t, 1);
                lds_store(0, T0{});
                gload(t + 4, T0{});
                IG_STAMP(t, 2);
                __syncthreads();
                IG_STAMP(t, 3);
                read_frags(T0{}, T0{});
                __builtin_amdgcn_sched_barrier(0);
                mma(T1{});
                __builtin_amdgcn_sched_barrier(0);
                lds_store(1, T1{});
                gload(t + 5, T1{});
                __syncthreads();
            }
            read_frags(T1{}, T1{});                 // step nk-4
            mma(T0{});
            lds_store(0, T0{});
            if (has_next) {
                setup_tile(next);
                gload(0, T0{});
            }
            __syncthreads();
            read_frags(T0{}, T0{});                 // step nk-3
            mma(T1{});
            lds_store(1, T1{});
            if (has_next) gload(1, T1{});
            __syncthreads();
            read_frags(T1{}, T1{});                 // step nk-2
            mma(T0{});
            __syncthreads();
            mma(T1{});                              // step nk-1
            __syncthreads();
        } else if constexpr (PF > 1) {
            // Step t fetches slice t+PF into the register set slice t came from, multiplies slice t, then
            // moves slice t+1 (fetched PF-1 steps earlier) into the other LDS buffer.  The steady-state
            // loop body is branch-free so that the compiler's wait counts let the newer fetches stay in
            // flight across the LDS store; the last PF steps fetch the next tile's slices 0..PF-1
            // (nk is a multiple of PF, so they land in sets 0..PF-1 again).
            for (int t = 0; t + PF < nk; t += PF) {
                static_for<0, PF>([&](auto st) {
                    constexpr int S = decltype(st)::value;
                    // the scheduling fences keep the compiler from hoisting the store's use of fetched
                    // registers (and with it a full wait) above the MFMAs
                    gload(t + S + PF, SetTag<S>{});
                    __builtin_amdgcn_sched_barrier(0);
                    compute();
                    __builtin_amdgcn_sched_barrier(0);
                    lds_store(cur ^ 1, SetTag<(S + 1) % PF>{});
                    __syncthreads();
                    cur ^= 1;
                });
            }
            static_for<0, PF>([&](auto st) {
                constexpr int S = decltype(st)::value;
                if (has_next) {
                    if constexpr (S == 0) setup_tile(next);
                    gload(S, SetTag<S>{});
                }
                compute();
                if constexpr (S + 1 < PF) lds_store(cur ^ 1, SetTag<S + 1>{});
                __syncthreads();
                cur ^= 1;
            });
        } else {
            for (int t = 0; t < nk; ++t) {
                const bool more = (t + 1) < nk;
                IG_STAMP(t, 0);
                if (more) {
                    gload(t + 1, SetTag<0>{});
                } else if (has_next) {   // prefetch the next tile's first slice; it lands during the epilogue
                    setup_tile(next);
                    gload(0, SetTag<0>{});
                }
                compute();
                IG_STAMP(t, 1);
                if (more) lds_store(cur ^ 1, SetTag<0>{});
                IG_STAMP(t, 2);
                __syncthreads();
                IG_STAMP(t, 3);
                cur ^= 1;
            }
        }

        // ---- epilogue: accumulators -> LDS (row-major tile) -> 16-B channel-contiguous stores ----
        // A lane holds column n = lr and rows (e&3) + 8*(e>>2) + 4*lh of each 32x32 block.
        const int n = en0 + en4 * 4;
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), bi4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (VEC) {
            if (n < p.Cout) {
                if (p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + n);
                if (p.bias) bi4 = *reinterpret_cast<const float4*>(p.bias + n);
            }
        } else {
            float* sp = reinterpret_cast<float*>(&sc4);
            float* bp = reinterpret_cast<float*>(&bi4);
            for (int u = 0; u < 4; ++u)
                if (n + u < p.Cout) {
                    if (p.scale) sp[u] = p.scale[n + u];
                    if (p.bias) bp[u] = p.bias[n + u];
                }
        }
        static_assert(KSPLIT * EP_R * EP_LD >= 2 * 4 * EP_RPI * BN4, "the epilogue tile image also holds the statistics scratch");
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
                                ep[(ks * EP_R + row) * EP_LD + wn * TN + j * 32 + lr] = acc[i][j][e];
                            }
                }
                __syncthreads();
                if (STATS && pass == 0) {
                    // the tile's first row (split-K planes added): a sample of every column to centre the moments on
                    st_shift = *reinterpret_cast<const float4*>(&ep[en4 * 4]);
                    if (tid < BN) my_shift = ep[tid];
#pragma unroll
                    for (int k2 = 1; k2 < KSPLIT; ++k2) {
                        const float4 o2 = *reinterpret_cast<const float4*>(&ep[(k2 * EP_R) * EP_LD + en4 * 4]);
                        st_shift.x += o2.x; st_shift.y += o2.y; st_shift.z += o2.z; st_shift.w += o2.w;
                        if (tid < BN) my_shift += ep[(k2 * EP_R) * EP_LD + tid];
                    }
                }
                for (int row = erow; row < EP_R && erow < EP_RPI; row += EP_RPI) {
                    const int m = em0 + r0 + row;
                    if constexpr (AMAX && VEC && (BN4 & (BN4 - 1)) == 0) {
                        {
                            // Row maximum over this tile's columns: the BN4 lanes that hold one row are contiguous
                            // in the wave, so a width-BN4 butterfly reduces them; first index wins ties.
                            float bv = -INFINITY;
                            int bi = 0x7fffffff;
                            if (m < p.M && n < p.Cout) {
                                float4 t4 = *reinterpret_cast<const float4*>(&ep[row * EP_LD + en4 * 4]);
#pragma unroll
                                for (int k2 = 1; k2 < KSPLIT; ++k2) {
                                    const float4 o2 = *reinterpret_cast<const float4*>(&ep[(k2 * EP_R + row) * EP_LD + en4 * 4]);
                                    t4.x += o2.x; t4.y += o2.y; t4.z += o2.z; t4.w += o2.w;
                                }
                                t4.x = act_c<ACT>(fmaf(t4.x, sc4.x, bi4.x)); t4.y = act_c<ACT>(fmaf(t4.y, sc4.y, bi4.y));
                                t4.z = act_c<ACT>(fmaf(t4.z, sc4.z, bi4.z)); t4.w = act_c<ACT>(fmaf(t4.w, sc4.w, bi4.w));
                                bv = t4.x; bi = n;
                                argmax_combine(bv, bi, t4.y, n + 1);
                                argmax_combine(bv, bi, t4.z, n + 2);
                                argmax_combine(bv, bi, t4.w, n + 3);
                            }
#pragma unroll
                            for (int o2 = BN4 / 2; o2 > 0; o2 >>= 1) {
                                const float ov = __shfl_xor(bv, o2, 64);
                                const int oi = __shfl_xor(bi, o2, 64);
                                argmax_combine(bv, bi, ov, oi);
                            }
                            if (en4 == 0 && m < p.M) {
                                p.amax_val[(size_t)m * p.tiles_n + en0 / BN] = bv;
                                p.amax_idx[(size_t)m * p.tiles_n + en0 / BN] = bi;
                            }
                        }
                    }
                    if (m >= p.M || n >= p.Cout) continue;
                    if (!p.y) continue;
                    float4 v = *reinterpret_cast<const float4*>(&ep[row * EP_LD + en4 * 4]);
#pragma unroll
                    for (int k2 = 1; k2 < KSPLIT; ++k2) {
                        const float4 o = *reinterpret_cast<const float4*>(&ep[(k2 * EP_R + row) * EP_LD + en4 * 4]);
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    if constexpr (STATS) stat_acc4(st_s1, st_s2, v, st_shift);
                    v.x = fmaf(v.x, sc4.x, bi4.x); v.y = fmaf(v.y, sc4.y, bi4.y);
                    v.z = fmaf(v.z, sc4.z, bi4.z); v.w = fmaf(v.w, sc4.w, bi4.w);
                    const size_t o = (size_t)m * p.ldy + n;
                    if constexpr (VEC) {
                        if (p.residual) {
                            const float4 rr = *reinterpret_cast<const float4*>(p.residual + o);
                            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                        }
                        v.x = act_c<ACT>(v.x); v.y = act_c<ACT>(v.y); v.z = act_c<ACT>(v.z); v.w = act_c<ACT>(v.w);
                        if (p.y) *reinterpret_cast<float4*>(p.y + o) = v;
                    } else {
                        const float* vv = reinterpret_cast<const float*>(&v);
                        for (int u = 0; u < 4; ++u)
                            if (n + u < p.Cout) {
                                float t2 = vv[u];
                                if (p.residual) t2 += p.residual[o + u];
                                p.y[o + u] = act_c<ACT>(t2);
                            }
                    }
                }
                if (STATS && pass == NPASS - 1) {
                    // column statistics of the result for the BatchNorm that follows: the row groups' shifted moments meet
                    // in LDS (the tile image is dead once every thread has read its rows) and one thread per column adds
                    // them in order
                    __syncthreads();
                    float4* sc = reinterpret_cast<float4*>(ep);                   // [EP_RPI][BN4][2]
                    if (erow < EP_RPI) {
                        sc[(erow * BN4 + en4) * 2] = st_s1;
                        sc[(erow * BN4 + en4) * 2 + 1] = st_s2;
                    }
                    __syncthreads();
                    if (tid < BN && en0 + tid < p.Cout) {
                        const int q4 = tid >> 2, e = tid & 3;
                        float a1 = 0.f, a2 = 0.f;
#pragma unroll
                        for (int r = 0; r < EP_RPI; ++r) {
                            a1 += reinterpret_cast<const float*>(&sc[(r * BN4 + q4) * 2])[e];
                            a2 += reinterpret_cast<const float*>(&sc[(r * BN4 + q4) * 2 + 1])[e];
                        }
                        const float nrows = (float)min(BM, p.M - em0);
                        float psum, pm2;
                        stat_from_shifted(nrows, my_shift, a1, a2, &psum, &pm2);
                        const size_t part = (size_t)em0 / BM;
                        p.stat_part[(part * 2 + 0) * p.Cout + en0 + tid] = psum;
                        p.stat_part[(part * 2 + 1) * p.Cout + en0 + tid] = pm2;
                    }
                }
                __syncthreads();   // the tile image is dead: the next pass / the next tile's slice 0 may overwrite it
            }
        });
        if (!has_next) break;
        tl = next;
    }
}

// Persistent grid: enough workgroups to fill every CU at the kernel's own occupancy, a multiple of 8.
// `cached` is a per-instantiation static owned by the caller (the value is idempotent, so a race between
// host threads is benign).
template <typename K>
static int persistent_grid(K kernel, int nblk, int& cached, int threads) {
    if (cached == 0) {
        int dev = 0, cus = 256, per_cu = 2;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        (void)hipGetLastError();
        cached = cus * per_cu;
    }
    if (nblk <= cached) return nblk;   // one tile per workgroup: any count works with the XCD remap
    return cached & ~7;                // several tiles per workgroup: the stride must keep tile % 8 fixed
}

// AMAX: the epilogue also emits per-tile row maxima (its own kernel instantiation, so that profiles show the
// vocabulary projection of greedy decoding under its own name).
template <int BM, int BN, int WM, int WN, int KSPLIT, int KU = 1, bool VEC = true, int PF = 1, bool AMAX = false, bool FRAG = false>
static int launch_cfg(ConvParams& p, hipStream_t st) {
    if (FRAG) {
        const int nk_ = (p.K + BK * KSPLIT * KU - 1) / (BK * KSPLIT * KU);
        if (nk_ < 4 || (nk_ & 1)) { set_error("conv_igemm: configuration needs an even number (>= 4) of K steps"); return ICAP_ERR_ARG; }
    }
    if (PF > 1 && (((p.K + BK * KSPLIT * KU - 1) / (BK * KSPLIT * KU)) % PF)) {
        set_error("conv_igemm: this tile configuration needs an even number of K steps");
        return ICAP_ERR_ARG;
    }
    const int tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    p.tiles_m = tiles_m;
    p.m_fast = (long long)p.Cout * p.K > (long long)p.B * p.H * p.W * p.Cin ? 1 : 0;
    const long long blocks = (long long)tiles_m * p.tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) {
        set_error("conv_igemm: grid of %lld workgroups out of range", blocks);
        return ICAP_ERR_LIMIT;
    }
    p.nblk = (int)blocks;
    const bool one = (p.KH == 1 && p.KW == 1 && p.pad_t == 0 && p.pad_l == 0);
    const dim3 block(64 * WM * WN * KSPLIT);
#define ICAP_LAUNCH(O_, G_, S_)                                                                      \
    do {                                                                                             \
        static int cached = 0;                                                                       \
        auto kern = conv_igemm_kernel<BM, BN, WM, WN, KSPLIT, KU, O_, G_, VEC, PF, AMAX, FRAG, S_>;                           \
        hipLaunchKernelGGL(kern, dim3((unsigned)persistent_grid(kern, p.nblk, cached, (int)block.x)), block, 0, st, p); \
    } while (0)
    if (p.stat_part) {
        if constexpr (VEC && !AMAX) {
            if (p.gate) { set_error("conv_igemm: the statistics epilogue takes no gate"); return ICAP_ERR_ARG; }
            if (one) ICAP_LAUNCH(true, false, true);
            else ICAP_LAUNCH(false, false, true);
        } else {
            set_error("conv_igemm: this configuration has no statistics epilogue");
            return ICAP_ERR_ARG;
        }
    }
    else if (one && p.gate) ICAP_LAUNCH(true, true, false);
    else if (one) ICAP_LAUNCH(true, false, false);
    else if (p.gate) ICAP_LAUNCH(false, true, false);
    else ICAP_LAUNCH(false, false, false);
#undef ICAP_LAUNCH
    return launch_status("conv_igemm");
}

// ---------------------------------------------------------------- tile configurations
// Every configuration computes the same result; they differ in how well they fill the chip for a given
// (rows, channels, K).  conv_igemm_f32 picks one by a shape heuristic unless conv_igemm_tune_f32 has
// measured the shape before, in which case the measured winner is used.
typedef int (*LaunchFn)(ConvParams&, hipStream_t);
struct TileCfg { int bm, bn, ksplit; LaunchFn fn; };
static const TileCfg kCfgs[] = {
    {128, 128, 1, launch_cfg<128, 128, 2, 2, 1>},   // 0
    {256, 64, 1, launch_cfg<256, 64, 4, 1, 1>},     // 1
    {256, 32, 1, launch_cfg<256, 32, 4, 1, 1>},     // 2
    {128, 32, 1, launch_cfg<128, 32, 4, 1, 1>},     // 3
    {64, 64, 1, launch_cfg<64, 64, 2, 2, 1>},       // 4
    {32, 32, 4, launch_cfg<32, 32, 1, 1, 4>},       // 5
    {128, 96, 1, launch_cfg<128, 96, 4, 1, 1>},     // 6
    {256, 96, 1, launch_cfg<256, 96, 4, 1, 1>},     // 7
    {128, 64, 1, launch_cfg<128, 64, 2, 2, 1>},     // 8
    {64, 128, 1, launch_cfg<64, 128, 2, 2, 1>},     // 9
    {256, 128, 1, launch_cfg<256, 128, 4, 1, 1>},   // 10
    {64, 64, 1, launch_cfg<64, 64, 2, 2, 1, 2>},    // 11: 32-wide K per wave and step (fewer, longer steps)
    {32, 32, 4, launch_cfg<32, 32, 1, 1, 4, 2>},    // 12
    {128, 64, 1, launch_cfg<128, 64, 2, 2, 1, 2>},  // 13
    {128, 128, 1, launch_cfg<128, 128, 2, 2, 1, 2>},// 14
    {64, 32, 2, launch_cfg<64, 32, 2, 1, 2, 2>},    // 15: 2 waves over rows x 2 over K
    // 8- and 16-wave workgroups: the K range of every step is split over 2 or 4 waves per output block, so a
    // problem with fewer tiles than the chip has CUs still keeps 2-4 waves on every SIMD of the CUs it uses.
    {64, 64, 2, launch_cfg<64, 64, 2, 2, 2, 1>},    // 16
    {64, 64, 2, launch_cfg<64, 64, 2, 2, 2, 2>},    // 17
    {64, 64, 4, launch_cfg<64, 64, 2, 2, 4, 1>},    // 18
    {128, 64, 2, launch_cfg<128, 64, 2, 2, 2, 2>},  // 19
    {32, 64, 4, launch_cfg<32, 64, 1, 2, 4, 2>},    // 20
    {32, 32, 8, launch_cfg<32, 32, 1, 1, 8, 1>},    // 21
    {64, 64, 1, launch_cfg<64, 64, 2, 2, 1, 2, true, 2>},    // 22: two K slices in flight (even step counts only)
    {32, 32, 4, launch_cfg<32, 32, 1, 1, 4, 1, true, 2>},    // 23
    {64, 64, 1, launch_cfg<64, 64, 2, 2, 1, 1, true, 2>},    // 24
    {64, 64, 1, launch_cfg<64, 64, 2, 2, 1, 1, true, 4>},    // 25: four K slices in flight
    {64, 64, 1, launch_cfg<64, 64, 2, 2, 1, 1, true, 8>},    // 26: eight
    {128, 64, 1, launch_cfg<128, 64, 2, 2, 1, 1, true, 4>},  // 27
    {32, 32, 4, launch_cfg<32, 32, 1, 1, 4, 1, true, 4>},    // 28
    // fragment-pipelined K loop (even step counts >= 4 only)
    {128, 128, 1, launch_cfg<128, 128, 2, 2, 1, 1, true, 1, false, true>},  // 29
    {64, 64, 1, launch_cfg<64, 64, 2, 2, 1, 1, true, 1, false, true>},      // 30
    {128, 64, 1, launch_cfg<128, 64, 2, 2, 1, 1, true, 1, false, true>},    // 31
    {64, 64, 1, launch_cfg<64, 64, 2, 2, 1, 2, true, 1, false, true>},      // 32
    {32, 32, 4, launch_cfg<32, 32, 1, 1, 4, 1, true, 1, false, true>},      // 33
    {256, 64, 1, launch_cfg<256, 64, 4, 1, 1, 1, true, 1, false, true>},    // 34
    {128, 96, 1, launch_cfg<128, 96, 4, 1, 1, 1, true, 1, false, true>},    // 35
};
constexpr int kNumCfgs = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));

static int heuristic_cfg(long long M, int Cout, long long K) {
    const long long tiles128 = ((M + 127) / 128) * ((Cout + 127) / 128);
    if (tiles128 < 256 && K >= 128) return 5;
    if (M <= 1024) return Cout <= 32 ? 3 : 4;
    if (Cout <= 32) return 2;
    if (Cout <= 64) return 1;
    return 0;
}

typedef std::tuple<int, int, int, int, int, int, int, int, int, int> ShapeKey;  // B,H,W,Cin,Cout,KH,KW,stride,Ho,Wo
static std::map<ShapeKey, int> g_tuned;
static std::mutex g_tuned_mu;
static bool g_use_tuned = true;      // false: conv_igemm_f32 ignores the table (verification reruns on the heuristic choice)

int conv_igemm_use_tuned(int flag) {
    std::lock_guard<std::mutex> lk(g_tuned_mu);
    const int prev = g_use_tuned ? 1 : 0;
    if (flag >= 0) g_use_tuned = flag != 0;
    return prev;
}

static int validate(const float* x, const float* w, const float* gate, float* y, int B, int H, int W, int Cin,
                    int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo) {
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
    return ICAP_OK;
}

static ConvParams make_params(const float* x, const float* w, const float* scale, const float* bias,
                              const float* residual, const float* gate, float* y, int B, int H, int W, int Cin,
                              int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act) {
    ConvParams p{x, w, scale, bias, residual, gate, y, B, H, W, Cin, Cout, KH, KW,
                 stride, pad_t, pad_l, Ho, Wo, act, B * Ho * Wo, KH * KW * Cin, Cout, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr};
    p.vec_ok = (Cout % 4 == 0) && aligned16(y) && (!residual || aligned16(residual)) &&
               (!scale || aligned16(scale)) && (!bias || aligned16(bias));
    return p;
}

int conv_igemm_f32(const float* x, const float* w, const float* scale, const float* bias,
                   const float* residual, const float* gate, float* y, int B, int H, int W, int Cin,
                   int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                   hipStream_t st) {
    const int rc = validate(x, w, gate, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo);
    if (rc != ICAP_OK) return rc;
    ConvParams p = make_params(x, w, scale, bias, residual, gate, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, act);
    // Outputs whose rows are not 16-byte sliceable (Cout % 4 != 0, e.g. the 81-symbol char vocabulary) take
    // one generic scalar-store configuration.
    if (!p.vec_ok) return launch_cfg<64, 64, 2, 2, 1, 1, false>(p, st);
    int cfg = -1;
    {
        std::lock_guard<std::mutex> lk(g_tuned_mu);
        auto it = g_use_tuned ? g_tuned.find(ShapeKey(B, H, W, Cin, Cout, KH, KW, stride, Ho, Wo)) : g_tuned.end();
        if (it != g_tuned.end()) cfg = it->second;
    }
    if (cfg < 0) cfg = heuristic_cfg(p.M, Cout, p.K);
    // tuning experiments only: ICAP_FORCE_CFG=<index> overrides the choice for every call
    static const int forced = [] { const char* e = getenv("ICAP_FORCE_CFG"); return e ? atoi(e) : -1; }();
    if (forced >= 0 && forced < kNumCfgs) {
        ConvParams pf = p;
        if (kCfgs[forced].fn(pf, st) == ICAP_OK) return ICAP_OK;   // a configuration that rejects the shape falls through
    }
    return kCfgs[cfg].fn(p, st);
}

// Measures every tile configuration on this exact problem (the output is computed for real each time) and
// remembers the fastest for later conv_igemm_f32 calls with the same shape.  Synchronises the stream: call it
// once per shape while warming up, never inside a captured or latency-critical region.
int conv_igemm_tune_f32(const float* x, const float* w, const float* scale, const float* bias,
                        const float* residual, const float* gate, float* y, int B, int H, int W, int Cin,
                        int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                        hipStream_t st) {
    const int rc = validate(x, w, gate, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo);
    if (rc != ICAP_OK) return rc;
    ConvParams p0 = make_params(x, w, scale, bias, residual, gate, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, act);
    if (!p0.vec_ok) return conv_igemm_f32(x, w, scale, bias, residual, gate, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, act, st);
    const ShapeKey key(B, H, W, Cin, Cout, KH, KW, stride, Ho, Wo);
    {
        std::lock_guard<std::mutex> lk(g_tuned_mu);
        if (g_tuned.count(key)) return kCfgs[g_tuned[key]].fn(p0, st);
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { set_error("conv_igemm_tune: event creation failed"); return ICAP_ERR_HIP_BASE; }
    int best = heuristic_cfg(p0.M, Cout, p0.K);
    float best_ms = 1e30f;
    for (int c = 0; c < kNumCfgs; ++c) {
        // skip tiles that are mostly padding
        if (kCfgs[c].bn >= 2 * Cout + 32 || (long long)kCfgs[c].bm >= 2 * (long long)p0.M + 64) continue;
        if (kCfgs[c].ksplit > 1 && p0.K < 128) continue;
        ConvParams p = p0;
        int r = kCfgs[c].fn(p, st);   // warm-up (also loads the code object)
        if (r != ICAP_OK) continue;
        // one timed launch sizes the repetition count so that short kernels are averaged over ~1 ms
        (void)hipEventRecord(e0, st);
        p = p0; kCfgs[c].fn(p, st);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) continue;
        float first_ms = 0.f;
        (void)hipEventElapsedTime(&first_ms, e0, e1);
        int reps = first_ms > 0.f ? (int)(1.0f / first_ms) : 3;
        reps = reps < 3 ? 3 : (reps > 40 ? 40 : reps);
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < reps; ++i) { p = p0; kCfgs[c].fn(p, st); }
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) continue;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= (float)reps;
        if (ms < best_ms) { best_ms = ms; best = c; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    {
        std::lock_guard<std::mutex> lk(g_tuned_mu);
        g_tuned[key] = best;
    }
    ConvParams p = p0;
    return kCfgs[best].fn(p, st);
}

// Linear layer whose epilogue also emits, per output row and 64-column tile, the maximum and its column
// (first index on ties): y (optional) [M][N], part_val / part_idx [M][linear_argmax_parts(N)].
// Residual is not supported here (an arg-max over logits never has one).
int linear_argmax_parts(int N) { return (N + 63) / 64; }

int linear_argmax_f32(const float* x, const float* w, const float* scale, const float* bias, float* y, float* part_val,
                      int* part_idx, int M, int K, int N, hipStream_t st) {
    if (!part_val || !part_idx) { set_error("linear_argmax: null partial buffers"); return ICAP_ERR_ARG; }
    float* yy = y ? y : reinterpret_cast<float*>(part_val);   // validate() needs a non-null, aligned output pointer
    const int rc = validate(x, w, nullptr, yy, M, 1, 1, K, N, 1, 1, 1, 0, 0, 1, 1);
    if (rc != ICAP_OK) return rc;
    ConvParams p = make_params(x, w, scale, bias, nullptr, nullptr, yy, M, 1, 1, K, N, 1, 1, 1, 0, 0, 1, 1, ACT_NONE);
    if (!p.vec_ok) { set_error("linear_argmax: N %% 4 and 16-byte aligned y / scale / bias required"); return ICAP_ERR_ALIGN; }
    p.y = y;
    p.amax_val = part_val;
    p.amax_idx = part_idx;
    const int ksteps = (K + BK - 1) / BK;
    static const int variant = [] { const char* e = getenv("ICAP_AMAX_VARIANT"); return e ? atoi(e) : 0; }();   // tuning experiments only
    if (variant == 1 && (ksteps & 1) == 0) return launch_cfg<64, 64, 2, 2, 1, 1, true, 2, true>(p, st);
    if (ksteps >= 4 && (ksteps & 1) == 0) return launch_cfg<64, 64, 2, 2, 1, 1, true, 1, true, true>(p, st);
    if ((ksteps & 1) == 0) return launch_cfg<64, 64, 2, 2, 1, 1, true, 2, true>(p, st);
    return launch_cfg<64, 64, 2, 2, 1, 2, true, 1, true>(p, st);
}

// Plain convolution / Linear (no scale, bias, residual, gate or activation) whose epilogue also leaves the column statistics
// of its result for a train-mode BatchNorm: stat_part [P][2][Cout] (sum, squared deviations), P = info[0] blocks of
// info[1] = BM rows each (one per row tile), for the tile configuration conv_igemm_f32 would pick for the shape.  The caller sizes
// stat_part for the worst case: conv_stats_max_parts(M) partials.
long long conv_stats_max_parts(long long M) { return (M + 31) / 32 + 8; }

int conv_igemm_stats_f32(const float* x, const float* w, float* y, int B, int H, int W, int Cin, int Cout, int KH, int KW,
                         int stride, int pad_t, int pad_l, int Ho, int Wo, float* stat_part, long long part_floats, int* info,
                         hipStream_t st) {
    const int rc = validate(x, w, nullptr, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo);
    if (rc != ICAP_OK) return rc;
    if (!stat_part || !info) { set_error("conv_igemm_stats: null statistics buffer"); return ICAP_ERR_ARG; }
    ConvParams p = make_params(x, w, nullptr, nullptr, nullptr, nullptr, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, ACT_NONE);
    if (!p.vec_ok) { set_error("conv_igemm_stats: Cout %% 4 == 0 and a 16-byte aligned y required"); return ICAP_ERR_ALIGN; }
    int cfg = -1;
    {
        std::lock_guard<std::mutex> lk(g_tuned_mu);
        auto it = g_use_tuned ? g_tuned.find(ShapeKey(B, H, W, Cin, Cout, KH, KW, stride, Ho, Wo)) : g_tuned.end();
        if (it != g_tuned.end()) cfg = it->second;
    }
    if (cfg < 0) cfg = heuristic_cfg(p.M, Cout, p.K);
    const int bm = kCfgs[cfg].bm, rows = bm;
    const long long parts = (long long)(p.M + bm - 1) / bm;         // one block of statistics per row tile
    if (parts * 2 * Cout > part_floats) {
        set_error("conv_igemm_stats: %lld partials of %d columns need %lld floats, the buffer holds %lld", parts, Cout, parts * 2 * Cout, part_floats);
        return ICAP_ERR_LIMIT;
    }
    info[0] = (int)parts;
    info[1] = rows;
    p.stat_part = stat_part;
    return kCfgs[cfg].fn(p, st);
}

// One named tile configuration on this problem (parity tests run every configuration the tuner may pick; a
// configuration that cannot run the shape - K-step parity of the pipelined forms - returns an argument error).
int conv_igemm_num_cfgs() { return kNumCfgs; }

int conv_igemm_cfg_f32(const float* x, const float* w, const float* scale, const float* bias,
                       const float* residual, const float* gate, float* y, int B, int H, int W, int Cin,
                       int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                       int cfg, hipStream_t st) {
    const int rc = validate(x, w, gate, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo);
    if (rc != ICAP_OK) return rc;
    if (cfg < 0 || cfg >= kNumCfgs) { set_error("conv_igemm: configuration %d out of range [0, %d)", cfg, kNumCfgs); return ICAP_ERR_ARG; }
    ConvParams p = make_params(x, w, scale, bias, residual, gate, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, act);
    if (!p.vec_ok) { set_error("conv_igemm: tile configurations need Cout %% 4 == 0 and 16-byte aligned y / residual / scale / bias"); return ICAP_ERR_ALIGN; }
    return kCfgs[cfg].fn(p, st);
}

int conv_igemm_tuned_cfg(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int Ho, int Wo) {
    std::lock_guard<std::mutex> lk(g_tuned_mu);
    auto it = g_tuned.find(ShapeKey(B, H, W, Cin, Cout, KH, KW, stride, Ho, Wo));
    return it == g_tuned.end() ? -1 : it->second;
}

}  // namespace icap
