// Weight-streaming kernels for the per-step decoder forward at small batch
// (1..8 captions decoded in lock-step).  At these sizes a step reads every weight
// once and does almost no arithmetic, so a kernel's time is launch latency plus a
// few dependent memory round trips.  The kernels are therefore built to
//   * put every weight load of a wave in flight before anything else happens
//     (rows are split over enough workgroups that a lane holds 2..8 16-byte loads),
//   * build their input vectors in LDS themselves (StepInput: token arg-max from
//     the previous vocabulary projection's partials, embedding-row gather,
//     residual add, LayerNorm), so a decode step is 2-3 launches, and
//   * fuse what follows the dot products: bias / activation, the LSTM or GRU cell
//     update, the vocabulary arg-max partials.
#include "decode.hpp"

namespace icap {

constexpr int GEMV_MAX_NB = 8;

template <int NB>
__device__ __forceinline__ void wave_sum_n(float (&a)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) a[b] = wave_sum(a[b]);
}

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sel(bool ok, float4 a) { return ok ? a : f4_zero(); }

constexpr int PART_STAGE = 4096;  // arg-max partials (all captions) a workgroup stages in LDS
constexpr int LN_PRE = 4;         // LayerNorm gamma/beta chunks prefetched per lane (covers K <= 1024)
constexpr int XU = 4;             // 16-byte vector elements a thread keeps in flight per staging pass

struct StageShared {
    float pv[PART_STAGE];
    int pi[PART_STAGE];
    int toks[GEMV_MAX_NB];
};

// ---- staging a plain [nb][K4] vector set (row stride ld floats) into LDS: the first XU*T elements are
// loaded into registers by vec_issue (no wait), vec_finish writes them and streams the rest.
template <int T>
__device__ __forceinline__ void vec_issue(const float* v, int ld, int n_valid, int K4, float4 (&r)[XU]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        if (u * T < n_valid) {
            const int i = min(tid + T * u, n_valid - 1);
            const int b = i / K4, k4 = i - b * K4;
            r[u] = *reinterpret_cast<const float4*>(v + (size_t)b * ld + (k4 << 2));
        }
    }
}

// dst[i] = r (+ ra) for the first pass, then the remaining elements straight from memory; rows >= nb are zeroed.
template <int NB, int T>
__device__ __forceinline__ void vec_finish(const float* v, int ld, const float* add, int ld_add, int nb, int K4, float4* dst,
                                           const float4 (&r)[XU], const float4 (&ra)[XU]) {
    const int tid = threadIdx.x;
    const int n_valid = nb * K4;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
        const int i = tid + T * u;
        if (i < n_valid) dst[i] = add ? f4_add(r[u], ra[u]) : r[u];
    }
    for (int base = XU * T; base < n_valid; base += XU * T) {
        float4 t[XU], ta[XU];
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = min(base + tid + T * u, n_valid - 1);
            const int b = i / K4, k4 = i - b * K4;
            t[u] = *reinterpret_cast<const float4*>(v + (size_t)b * ld + (k4 << 2));
            if (add) ta[u] = *reinterpret_cast<const float4*>(add + (size_t)b * ld_add + (k4 << 2));
        }
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int i = base + tid + T * u;
            if (i < n_valid) dst[i] = add ? f4_add(t[u], ta[u]) : t[u];
        }
    }
    for (int i = n_valid + tid; i < NB * K4; i += T) dst[i] = f4_zero();
}

// Registers a thread carries between the two halves of input staging.
template <int T>
struct StageRegs {
    float4 g[LN_PRE], be[LN_PRE];
    float lv[PART_STAGE / T];
    int li[PART_STAGE / T];
    float4 xv[XU], av[XU];
};

// First half: issue every global load whose address is known at kernel start
// (LayerNorm parameters, arg-max partials, a plain input vector) and wait for none.
// Called right after the weight loads, so that one memory round trip covers them all.
template <int T>
__device__ __forceinline__ void stage_issue(const StepInput& s, int nb, int K, StageRegs<T>& r) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int K4 = K >> 2;
    if (s.ln_g) {
#pragma unroll
        for (int i = 0; i < LN_PRE; ++i) {
            const int kc = min(lane + 64 * i, K4 - 1);
            r.g[i] = reinterpret_cast<const float4*>(s.ln_g)[kc];
            r.be[i] = reinterpret_cast<const float4*>(s.ln_b)[kc];
        }
    }
    if (s.table) {
        const int total = nb * s.nparts;
        if (s.part_val && total <= PART_STAGE) {
#pragma unroll
            for (int u = 0; u < PART_STAGE / T; ++u) {
                if (u * T < total) {
                    const int e = min(tid + T * u, total - 1);
                    r.lv[u] = s.part_val[e];
                    r.li[u] = s.part_idx[e];
                }
            }
        }
    } else {
        vec_issue<T>(s.x, s.ldx, nb * K4, K4, r.xv);
        if (s.add) vec_issue<T>(s.add, s.ld_add, nb * K4, K4, r.av);
    }
}

// Second half: builds the [NB][K/4] input vectors in LDS (see StepInput).  Rows b >= nb
// are zero.  Ends with a workgroup barrier; all T threads must call it.  The only
// dependent memory round trip left in here is token -> embedding row.
template <int NB, int T>
__device__ __forceinline__ void stage_finish(const StepInput& s, int nb, int K, float4* xs, StageShared& sh, const StageRegs<T>& r) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int WAVES = T / 64;
    const int K4 = K >> 2, E4 = s.E >> 2;
    if (s.table) {
        if (s.part_val) {
            const int total = nb * s.nparts;
            if (total <= PART_STAGE) {
#pragma unroll
                for (int u = 0; u < PART_STAGE / T; ++u) {
                    const int e = tid + T * u;
                    if (e < total) { sh.pv[e] = r.lv[u]; sh.pi[e] = r.li[u]; }
                }
                __syncthreads();
                for (int b = wave; b < nb; b += WAVES) {
                    float v = -INFINITY;
                    int i = 0x7fffffff;
                    for (int p = lane; p < s.nparts; p += 64) argmax_combine(v, i, sh.pv[b * s.nparts + p], sh.pi[b * s.nparts + p]);
                    wave_argmax(v, i);
                    if (lane == 0) sh.toks[b] = (i == 0x7fffffff) ? 0 : i;  // all-NaN row
                }
            } else {
                for (int b = wave; b < nb; b += WAVES) {
                    float v = -INFINITY;
                    int i = 0x7fffffff;
                    for (int p = lane; p < s.nparts; p += 64)
                        argmax_combine(v, i, s.part_val[(size_t)b * s.nparts + p], s.part_idx[(size_t)b * s.nparts + p]);
                    wave_argmax(v, i);
                    if (lane == 0) sh.toks[b] = (i == 0x7fffffff) ? 0 : i;
                }
            }
        } else if (tid < nb) {
            sh.toks[tid] = (int)s.tok_in[tid];
        }
        __syncthreads();
        if (blockIdx.x == 0 && s.ids_out && tid < nb) s.ids_out[(size_t)tid * s.ids_stride] = sh.toks[tid];
        // v[b] = [ table[tok[b]] (+ add[b]) | tail[b] ], XU loads per thread in flight
        const int n_valid = nb * K4;
        for (int base = 0; base < n_valid; base += XU * T) {
            float4 t[XU], ta[XU];
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                if (base + u * T < n_valid) {
                    const int i = min(base + tid + T * u, n_valid - 1);
                    const int b = i / K4, k4 = i - b * K4;
                    // whatever the token, the row read stays inside the table
                    const int tok = min(max(sh.toks[b], 0), s.table_rows - 1);
                    const bool head = k4 < E4;
                    const float* src = head ? s.table + (size_t)tok * s.E + (k4 << 2)
                                            : s.tail + (size_t)b * s.ld_tail + ((k4 - E4) << 2);
                    t[u] = *reinterpret_cast<const float4*>(src);
                    if (s.add) ta[u] = f4_sel(head, *reinterpret_cast<const float4*>(s.add + (size_t)b * s.ld_add + (min(k4, E4 - 1) << 2)));
                }
            }
#pragma unroll
            for (int u = 0; u < XU; ++u) {
                const int i = base + tid + T * u;
                if (i < n_valid) xs[i] = s.add ? f4_add(t[u], ta[u]) : t[u];
            }
        }
        for (int i = n_valid + tid; i < NB * K4; i += T) xs[i] = f4_zero();
    } else {
        vec_finish<NB, T>(s.x, s.ldx, s.add, s.ld_add, nb, K4, xs, r.xv, r.av);
    }
    if (s.ln_g) {
        __syncthreads();
        for (int b = wave; b < nb; b += WAVES) {
            float4* row = xs + b * K4;
            float sum = 0.f;
            for (int k4 = lane; k4 < K4; k4 += 64) { const float4 v = row[k4]; sum += (v.x + v.y) + (v.z + v.w); }
            const float mean = wave_sum(sum) / (float)K;
            float sq = 0.f;
            for (int k4 = lane; k4 < K4; k4 += 64) {
                const float4 v = row[k4];
                const float a = v.x - mean, c = v.y - mean, d = v.z - mean, e = v.w - mean;
                sq += (a * a + c * c) + (d * d + e * e);
            }
            const float rstd = rsqrtf(wave_sum(sq) / (float)K + s.eps);
#pragma unroll
            for (int i = 0; i < LN_PRE; ++i) {
                const int k4 = lane + 64 * i;
                if (k4 < K4) {
                    const float4 v = row[k4];
                    row[k4] = make_float4((v.x - mean) * rstd * r.g[i].x + r.be[i].x, (v.y - mean) * rstd * r.g[i].y + r.be[i].y,
                                          (v.z - mean) * rstd * r.g[i].z + r.be[i].z, (v.w - mean) * rstd * r.g[i].w + r.be[i].w);
                }
            }
            for (int k4 = lane + 64 * LN_PRE; k4 < K4; k4 += 64) {
                const float4 v = row[k4];
                const float4 gg = reinterpret_cast<const float4*>(s.ln_g)[k4];
                const float4 bb = reinterpret_cast<const float4*>(s.ln_b)[k4];
                row[k4] = make_float4((v.x - mean) * rstd * gg.x + bb.x, (v.y - mean) * rstd * gg.y + bb.y,
                                      (v.z - mean) * rstd * gg.z + bb.z, (v.w - mean) * rstd * gg.w + bb.w);
            }
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && s.x_out) {
        for (int i = tid; i < nb * K4; i += T) {
            const int b = i / K4, k4 = i - b * K4;
            *reinterpret_cast<float4*>(s.x_out + (size_t)b * s.ld_xout + (k4 << 2)) = xs[i];
        }
    }
}

// One weight row segment per lane: IT 16-byte loads, issued unconditionally from
// clamped addresses (a conditional load makes the compiler wait at every one).
template <int IT>
__device__ __forceinline__ void row_load(float4 (&w)[IT], const float4* row, int K4, int k0, int stride) {
#pragma unroll
    for (int i = 0; i < IT; ++i) w[i] = row[min(k0 + i * stride, K4 - 1)];
}

// Dot products of one 16-byte weight segment per i with the staged vectors.  KU counts the row's 16-byte units:
// K/4 for fp32 weights, K/8 for bf16 weights (W16: a unit holds eight weights, widened exactly by a 16-bit shift,
// and meets two consecutive float4 of the vector); K4 = K/4 is the vectors' row pitch.
template <int IT, int NB, bool W16>
__device__ __forceinline__ void row_dot(const float4 (&w)[IT], const float4* vec, int K4, int KU, int k0, int stride, float (&acc)[NB]) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int ku = k0 + i * stride;
        const float4 wv = f4_sel(ku < KU, w[i]);
        const int kc = min(ku, KU - 1);
        if constexpr (!W16) {
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = dot4(wv, vec[b * K4 + kc], acc[b]);
        } else {
            const unsigned u0 = __float_as_uint(wv.x), u1 = __float_as_uint(wv.y), u2 = __float_as_uint(wv.z), u3 = __float_as_uint(wv.w);
            const float4 lo = make_float4(__uint_as_float(u0 << 16), __uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 << 16),
                                          __uint_as_float(u1 & 0xffff0000u));
            const float4 hi = make_float4(__uint_as_float(u2 << 16), __uint_as_float(u2 & 0xffff0000u), __uint_as_float(u3 << 16),
                                          __uint_as_float(u3 & 0xffff0000u));
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc[b] = dot4(lo, vec[b * K4 + 2 * kc], acc[b]);
                acc[b] = dot4(hi, vec[b * K4 + 2 * kc + 1], acc[b]);
            }
        }
    }
}

// ---------------------------------------------------------------- GEMV (+ optional arg-max partials)
// y[b][n] = act((W[n,:] . v[b,:]) * scale[n] + bias[n]),  b < nb, W row-major [N][K], K % 4 == 0, v from StepInput.
// A workgroup of T/64 waves owns RB = R * (T/64) / KW rows; KW waves share a row's K range.
// With part_val the workgroup also writes its best (value, index) per b to
// part_val/part_idx[b][blockIdx.x]; y may then be null.
template <int NB, int T, int R, int KW, int IT, bool W16>
__global__ __launch_bounds__(T) void gemv_kernel(const void* __restrict__ W, const float* __restrict__ scale,
                                                 const float* __restrict__ bias, const StepInput in, int nb,
                                                 float* __restrict__ y, int ldy, int N, int K, int act,
                                                 float* __restrict__ part_val, int* __restrict__ part_idx) {
    constexpr int WR = (T / 64) / KW, RB = R * WR;
    static_assert(RB * NB <= T, "one epilogue thread per (row, caption)");
    extern __shared__ __attribute__((aligned(16))) float4 xs[];  // [NB][K/4]
    __shared__ float red[KW][RB][NB];
    __shared__ float fin[RB][NB];
    __shared__ StageShared sh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave / KW, kp = wave % KW;
    const int K4 = K >> 2, KU = W16 ? K >> 3 : K4;  // 16-byte units of a weight row
    constexpr int stride = 64 * KW;
    const int k0 = kp * 64 + lane;
    const float4* W4 = reinterpret_cast<const float4*>(W);
    const float4* rows[R];
#pragma unroll
    for (int r = 0; r < R; ++r) rows[r] = W4 + (size_t)min(blockIdx.x * RB + wr * R + r, N - 1) * KU;

    float4 wv[R][IT];
#pragma unroll
    for (int r = 0; r < R; ++r) row_load<IT>(wv[r], rows[r], KU, k0, stride);
    // epilogue operands of the threads that finish a (row, caption) pair, fetched up front
    const int er = tid / NB, eb = tid - er * NB;
    const int en = blockIdx.x * RB + er;
    const bool e_ok = tid < RB * NB && en < N && eb < nb;
    const float e_scale = (e_ok && scale) ? scale[en] : 1.f;
    const float e_bias = (e_ok && bias) ? bias[en] : 0.f;
    StageRegs<T> sr;
    stage_issue<T>(in, nb, K, sr);

    stage_finish<NB, T>(in, nb, K, xs, sh, sr);

    float acc[R][NB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
    for (int base = 0;;) {
#pragma unroll
        for (int r = 0; r < R; ++r) row_dot<IT, NB, W16>(wv[r], xs, K4, KU, base + k0, stride, acc[r]);
        base += IT * stride;
        if (base >= KU) break;
#pragma unroll
        for (int r = 0; r < R; ++r) row_load<IT>(wv[r], rows[r], KU, base + k0, stride);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        wave_sum_n<NB>(acc[r]);
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) red[kp][wr * R + r][b] = acc[r][b];
        }
    }
    __syncthreads();
    if (tid < RB * NB) {
        float v = -INFINITY;
        if (e_ok) {
            float s = red[0][er][eb];
#pragma unroll
            for (int k = 1; k < KW; ++k) s += red[k][er][eb];
            v = act_apply(fmaf(s, e_scale, e_bias), act);
            if (y) y[(size_t)eb * ldy + en] = v;
        }
        fin[er][eb] = v;
    }
    if (part_val) {
        __syncthreads();
        if (tid < nb) {
            float v = -INFINITY;
            int i = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int n = blockIdx.x * RB + r;
                if (n < N) argmax_combine(v, i, fin[r][tid], n);
            }
            part_val[(size_t)tid * gridDim.x + blockIdx.x] = v;
            part_idx[(size_t)tid * gridDim.x + blockIdx.x] = i;
        }
    }
}

// Rows per workgroup, a function of N alone so that callers can size the partial buffers:
//   N <= 1024  one row, its K range split over the 4 waves (few rows, long K: fc2-like)
//   N <= 4096  4 rows, one per wave
//   larger     16 rows, four per wave (vocabulary projections: many short rows, and few
//              arg-max partials for the next kernel to reduce)
static int gemv_rows_per_block(int N) { return N <= 1024 ? 1 : (N <= 4096 ? 4 : 16); }

int gemv_blocks(int N) {
    const int rb = gemv_rows_per_block(N);
    return (N + rb - 1) / rb;
}

template <int NB, int T, int R, int KW, bool W16>
static void gemv_launch_it(int KU, int blocks, size_t lds, hipStream_t st, const void* W, const float* scale,
                           const float* bias, const StepInput& in, int nb, float* y, int ldy, int N, int K, int act,
                           float* pv, int* pi) {
    if (KU > 128 * KW)
        hipLaunchKernelGGL((gemv_kernel<NB, T, R, KW, 4, W16>), dim3(blocks), dim3(T), lds, st, W, scale, bias, in, nb, y, ldy, N, K, act, pv, pi);
    else if (!W16 || KU > 64 * KW)
        hipLaunchKernelGGL((gemv_kernel<NB, T, R, KW, 2, W16>), dim3(blocks), dim3(T), lds, st, W, scale, bias, in, nb, y, ldy, N, K, act, pv, pi);
    else
        hipLaunchKernelGGL((gemv_kernel<NB, T, R, KW, 1, W16>), dim3(blocks), dim3(T), lds, st, W, scale, bias, in, nb, y, ldy, N, K, act, pv, pi);
}

template <int NB, bool W16>
static void gemv_launch(int rb, int KU, int blocks, size_t lds, hipStream_t st, const void* W, const float* scale,
                        const float* bias, const StepInput& in, int nb, float* y, int ldy, int N, int K, int act, float* pv,
                        int* pi) {
    if (rb == 1) gemv_launch_it<NB, 256, 1, 4, W16>(KU, blocks, lds, st, W, scale, bias, in, nb, y, ldy, N, K, act, pv, pi);
    else if (rb == 4) gemv_launch_it<NB, 256, 1, 1, W16>(KU, blocks, lds, st, W, scale, bias, in, nb, y, ldy, N, K, act, pv, pi);
    else gemv_launch_it<NB, 256, 4, 1, W16>(KU, blocks, lds, st, W, scale, bias, in, nb, y, ldy, N, K, act, pv, pi);
}

template <bool W16>
static void gemv_dispatch(int nbt, int rb, int KU, int blocks, size_t lds, hipStream_t st, const void* W, const float* scale,
                          const float* bias, const StepInput& in, int nb, float* y, int ldy, int N, int K, int act, float* pv,
                          int* pi) {
    switch (nbt) {
        case 1: gemv_launch<1, W16>(rb, KU, blocks, lds, st, W, scale, bias, in, nb, y, ldy, N, K, act, pv, pi); break;
        case 2: gemv_launch<2, W16>(rb, KU, blocks, lds, st, W, scale, bias, in, nb, y, ldy, N, K, act, pv, pi); break;
        case 4: gemv_launch<4, W16>(rb, KU, blocks, lds, st, W, scale, bias, in, nb, y, ldy, N, K, act, pv, pi); break;
        default: gemv_launch<8, W16>(rb, KU, blocks, lds, st, W, scale, bias, in, nb, y, ldy, N, K, act, pv, pi); break;
    }
}

static int check_input(const char* what, const StepInput& in, int NB, int K) {
    if (in.table) {
        if (in.E <= 0 || in.E > K || in.table_rows <= 0 || (!in.part_val && !in.tok_in) || (in.part_val && (!in.part_idx || in.nparts <= 0)) ||
            (in.E < K && !in.tail) || (in.ids_out && in.ids_stride <= 0)) {
            set_error("%s: bad token input descriptor", what);
            return ICAP_ERR_ARG;
        }
        if (in.E % 4 || !aligned16(in.table) || (in.tail && (in.ld_tail % 4 || !aligned16(in.tail)))) {
            set_error("%s: token input needs E, ld_tail %% 4 and 16-byte alignment", what);
            return ICAP_ERR_ALIGN;
        }
    } else {
        if (!in.x || in.ldx < K) { set_error("%s: bad input vector", what); return ICAP_ERR_ARG; }
        if (in.ldx % 4 || !aligned16(in.x)) { set_error("%s: ldx %% 4 and 16-byte alignment required", what); return ICAP_ERR_ALIGN; }
    }
    if (in.add && (in.ld_add % 4 || !aligned16(in.add))) { set_error("%s: ld_add %% 4 and 16-byte alignment required", what); return ICAP_ERR_ALIGN; }
    if (in.ln_g && (!in.ln_b || !aligned16(in.ln_g) || !aligned16(in.ln_b))) { set_error("%s: bad LayerNorm parameters", what); return ICAP_ERR_ARG; }
    if (in.x_out && (in.ld_xout % 4 || in.ld_xout < K || !aligned16(in.x_out))) { set_error("%s: bad x_out", what); return ICAP_ERR_ALIGN; }
    (void)NB;
    return ICAP_OK;
}

// w16: W holds bf16 weights (round to nearest even of the fp32 matrix, same [N][K] layout); products and sums stay fp32.
int gemv_step_f32(const void* W, const float* scale, const float* bias, const StepInput& in, float* y, int ldy, int NB,
                  int N, int K, int act, float* part_val, int* part_idx, hipStream_t st, bool w16) {
    const bool argmax = part_val != nullptr;
    if (!W || (!y && !argmax) || (argmax && !part_idx) || NB <= 0 || N <= 0 || K <= 0) {
        set_error("gemv: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (NB > GEMV_MAX_NB) { set_error("gemv: batch %d exceeds %d (use the GEMM path)", NB, GEMV_MAX_NB); return ICAP_ERR_LIMIT; }
    if (K % 4 || !aligned16(W)) { set_error("gemv: K %% 4 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    if (w16 && K % 8) { set_error("gemv: bf16 weight rows need K=%d to be a multiple of 8", K); return ICAP_ERR_ALIGN; }
    const int rc = check_input("gemv", in, NB, K);
    if (rc != ICAP_OK) return rc;
    const int nbt = NB <= 1 ? 1 : (NB <= 2 ? 2 : (NB <= 4 ? 4 : 8));
    const size_t lds = (size_t)nbt * K * sizeof(float);
    if (lds > 96 * 1024) { set_error("gemv: NB*K=%d floats do not fit the LDS budget", nbt * K); return ICAP_ERR_LIMIT; }
    const int rb = gemv_rows_per_block(N), blocks = gemv_blocks(N);
    if (w16) gemv_dispatch<true>(nbt, rb, K / 8, blocks, lds, st, W, scale, bias, in, NB, y, ldy, N, K, act, part_val, part_idx);
    else gemv_dispatch<false>(nbt, rb, K / 4, blocks, lds, st, W, scale, bias, in, NB, y, ldy, N, K, act, part_val, part_idx);
    return launch_status("gemv");
}

int gemv_f32(const float* W, const float* scale, const float* bias, const float* x, int ldx, float* y, int ldy, int NB,
             int N, int K, int act, float* part_val, int* part_idx, hipStream_t st) {
    return gemv_step_f32(W, scale, bias, plain_input(x, ldx), y, ldy, NB, N, K, act, part_val, part_idx, st, false);
}

// ---------------------------------------------------------------- arg-max finalize (+ next-token gather)
// Reduces the per-workgroup partials of a vocabulary GEMV to one token per caption,
// stores it at ids[b*ids_stride], and optionally gathers the next step's input row
// next_x[b][:] = table[token][:] (+ add[b][:]).  One workgroup per caption.
__global__ __launch_bounds__(256) void argmax_finalize_kernel(const float* __restrict__ part_val,
                                                              const int* __restrict__ part_idx, int nparts,
                                                              long long* __restrict__ ids, int ids_stride,
                                                              const float* __restrict__ table,
                                                              const float* __restrict__ add,
                                                              float* __restrict__ next_x, int ldx, int E) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int tok;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float v = -INFINITY;
    int i = 0x7fffffff;
    for (int p = tid; p < nparts; p += 256) argmax_combine(v, i, part_val[(size_t)b * nparts + p], part_idx[(size_t)b * nparts + p]);
    wave_argmax(v, i);
    if (lane == 0) { sv[wave] = v; si[wave] = i; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) argmax_combine(v, i, sv[w], si[w]);
        if (i == 0x7fffffff) i = 0;  // all-NaN row
        tok = i;
        ids[(size_t)b * ids_stride] = i;
    }
    __syncthreads();
    if (table) {
        const float* src = table + (size_t)tok * E;
        for (int e = tid; e < E; e += 256) {
            float t = src[e];
            if (add) t += add[(size_t)b * E + e];
            next_x[(size_t)b * ldx + e] = t;
        }
    }
}

int argmax_finalize_f32(const float* part_val, const int* part_idx, int nparts, long long* ids, int ids_stride,
                        const float* table, const float* add, float* next_x, int ldx, int E, int NB, hipStream_t st) {
    if (!part_val || !part_idx || !ids || nparts <= 0 || NB <= 0 || ids_stride <= 0 || (table && (!next_x || E <= 0 || ldx < E))) {
        set_error("argmax_finalize: bad arguments");
        return ICAP_ERR_ARG;
    }
    hipLaunchKernelGGL(argmax_finalize_kernel, dim3(NB), dim3(256), 0, st, part_val, part_idx, nparts, ids, ids_stride, table, add, next_x, ldx, E);
    return launch_status("argmax_finalize");
}

// ---------------------------------------------------------------- fused LSTM / GRU cell
// One workgroup per hidden unit j, one wave per gate:
//   LSTM  wave g in (i, f, g, o): rows g*H+j of W_ih and W_hh, one accumulator;
//   GRU   wave 0: r (both rows), wave 1: z (both rows), wave 2: W_in x, wave 3: W_hn h
//         (n = tanh(W_in x + b_in + r * (W_hn h + b_hn)) needs the two halves apart).
// The gate pre-activations meet in LDS and lanes 0..nb-1 apply the cell update.
// x comes from StepInput ([NB][I]); h_in/c_in/h_out/c_out are [NB][H].
template <int NB, int IT, bool LSTM, bool W16>
__global__ __launch_bounds__(256) void rnn_cell_kernel(const void* __restrict__ w_ih, const void* __restrict__ w_hh,
                                                       const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                       const StepInput in, int nb, const float* __restrict__ h_in,
                                                       const float* __restrict__ c_in, float* __restrict__ h_out,
                                                       float* __restrict__ c_out, int I, int H) {
    extern __shared__ __attribute__((aligned(16))) float4 vs[];  // x: [NB][I/4] then h: [NB][H/4]
    __shared__ float pre[4][NB];
    __shared__ StageShared sh;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int I4 = I >> 2, H4 = H >> 2;
    const int IU = W16 ? I >> 3 : I4, HU = W16 ? H >> 3 : H4;  // 16-byte units of a weight row
    const int j = blockIdx.x;
    float4* xs = vs;
    float4* hs = vs + NB * I4;
    const bool has_a = LSTM || wave < 3, has_b = LSTM || wave != 2;
    const int gate = LSTM ? wave : min(wave, 2);
    const float4* row_a = reinterpret_cast<const float4*>(w_ih) + (size_t)(gate * H + j) * IU;
    const float4* row_b = reinterpret_cast<const float4*>(w_hh) + (size_t)(gate * H + j) * HU;
    constexpr int stride = 64;
    float4 wa[IT], wb[IT];
    if (has_a) row_load<IT>(wa, row_a, IU, lane, stride);
    if (has_b) row_load<IT>(wb, row_b, HU, lane, stride);
    float4 hv[XU];
    vec_issue<256>(h_in, H, nb * H4, H4, hv);
    // operands of the cell update (threads 0..nb-1), fetched up front
    float bsum[4] = {0.f, 0.f, 0.f, 0.f}, b_hn = 0.f, c_prev = 0.f, h_prev = 0.f;
    if (tid < nb) {
        const size_t o = (size_t)tid * H + j;
        if (LSTM) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bsum[q] = b_ih[q * H + j] + b_hh[q * H + j];
            c_prev = c_in[o];
        } else {
            bsum[0] = b_ih[j] + b_hh[j];
            bsum[1] = b_ih[H + j] + b_hh[H + j];
            bsum[2] = b_ih[2 * H + j];
            b_hn = b_hh[2 * H + j];
            h_prev = h_in[o];
        }
    }
    StageRegs<256> sr;
    stage_issue<256>(in, nb, I, sr);

    vec_finish<NB, 256>(h_in, H, nullptr, 0, nb, H4, hs, hv, hv);
    stage_finish<NB, 256>(in, nb, I, xs, sh, sr);

    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.f;
    if (has_a) {
        for (int base = 0;;) {
            row_dot<IT, NB, W16>(wa, xs, I4, IU, base + lane, stride, acc);
            base += IT * stride;
            if (base >= IU) break;
            row_load<IT>(wa, row_a, IU, base + lane, stride);
        }
    }
    if (has_b) {
        for (int base = 0;;) {
            row_dot<IT, NB, W16>(wb, hs, H4, HU, base + lane, stride, acc);
            base += IT * stride;
            if (base >= HU) break;
            row_load<IT>(wb, row_b, HU, base + lane, stride);
        }
    }
    wave_sum_n<NB>(acc);
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < NB; ++b) pre[wave][b] = acc[b];
    }
    __syncthreads();
    if (tid < nb) {
        const int b = tid;
        const size_t o = (size_t)b * H + j;
        if (LSTM) {
            const float ig = sigmoidf_(pre[0][b] + bsum[0]);
            const float fg = sigmoidf_(pre[1][b] + bsum[1]);
            const float gg = tanhf(pre[2][b] + bsum[2]);
            const float og = sigmoidf_(pre[3][b] + bsum[3]);
            const float c = fg * c_prev + ig * gg;
            c_out[o] = c;
            h_out[o] = og * tanhf(c);
        } else {
            const float r = sigmoidf_(pre[0][b] + bsum[0]);
            const float z = sigmoidf_(pre[1][b] + bsum[1]);
            const float n = tanhf((pre[2][b] + bsum[2]) + r * (pre[3][b] + b_hn));
            h_out[o] = (1.f - z) * n + z * h_prev;
        }
    }
}

template <int NB, bool LSTM, bool W16>
static void cell_launch(int it, int H, size_t lds, hipStream_t st, const void* w_ih, const void* w_hh, const float* b_ih,
                        const float* b_hh, const StepInput& in, int nb, const float* h_in, const float* c_in, float* h_out,
                        float* c_out, int I) {
    if (it > 2)
        hipLaunchKernelGGL((rnn_cell_kernel<NB, 4, LSTM, W16>), dim3(H), dim3(256), lds, st, w_ih, w_hh, b_ih, b_hh, in, nb, h_in, c_in, h_out, c_out, I, H);
    else if (!W16 || it > 1)
        hipLaunchKernelGGL((rnn_cell_kernel<NB, 2, LSTM, W16>), dim3(H), dim3(256), lds, st, w_ih, w_hh, b_ih, b_hh, in, nb, h_in, c_in, h_out, c_out, I, H);
    else
        hipLaunchKernelGGL((rnn_cell_kernel<NB, 1, LSTM, W16>), dim3(H), dim3(256), lds, st, w_ih, w_hh, b_ih, b_hh, in, nb, h_in, c_in, h_out, c_out, I, H);
}

template <bool LSTM, bool W16>
static void cell_dispatch(int nbt, int it, int H, size_t lds, hipStream_t st, const void* w_ih, const void* w_hh,
                          const float* b_ih, const float* b_hh, const StepInput& in, int nb, const float* h_in,
                          const float* c_in, float* h_out, float* c_out, int I) {
    switch (nbt) {
        case 1: cell_launch<1, LSTM, W16>(it, H, lds, st, w_ih, w_hh, b_ih, b_hh, in, nb, h_in, c_in, h_out, c_out, I); break;
        case 2: cell_launch<2, LSTM, W16>(it, H, lds, st, w_ih, w_hh, b_ih, b_hh, in, nb, h_in, c_in, h_out, c_out, I); break;
        case 4: cell_launch<4, LSTM, W16>(it, H, lds, st, w_ih, w_hh, b_ih, b_hh, in, nb, h_in, c_in, h_out, c_out, I); break;
        default: cell_launch<8, LSTM, W16>(it, H, lds, st, w_ih, w_hh, b_ih, b_hh, in, nb, h_in, c_in, h_out, c_out, I); break;
    }
}

// w16: w_ih / w_hh hold bf16 weights (same layout); biases, state and sums stay fp32.
int rnn_cell_step_f32(bool lstm, const void* w_ih, const void* w_hh, const float* b_ih, const float* b_hh,
                      const StepInput& in, const float* h_in, const float* c_in, float* h_out, float* c_out, int NB, int I,
                      int H, hipStream_t st, bool w16) {
    const char* what = lstm ? "lstm_cell" : "gru_cell";
    if (!w_ih || !w_hh || !b_ih || !b_hh || !h_in || !h_out || (lstm && (!c_in || !c_out)) || NB <= 0 || I <= 0 || H <= 0) {
        set_error("%s: bad arguments", what);
        return ICAP_ERR_ARG;
    }
    if (h_in == h_out) { set_error("%s: h_out must not alias h_in", what); return ICAP_ERR_ARG; }
    if (NB > GEMV_MAX_NB) { set_error("%s: batch %d exceeds %d (use the GEMM path)", what, NB, GEMV_MAX_NB); return ICAP_ERR_LIMIT; }
    if (I % 4 || H % 4 || !aligned16(w_ih) || !aligned16(w_hh) || !aligned16(h_in)) {
        set_error("%s: I, H %% 4 and 16-byte alignment required", what);
        return ICAP_ERR_ALIGN;
    }
    if (w16 && (I % 8 || H % 8)) { set_error("%s: bf16 weight rows need I=%d and H=%d to be multiples of 8", what, I, H); return ICAP_ERR_ALIGN; }
    const int rc = check_input(what, in, NB, I);
    if (rc != ICAP_OK) return rc;
    const int nbt = NB <= 1 ? 1 : (NB <= 2 ? 2 : (NB <= 4 ? 4 : 8));
    const size_t lds = (size_t)nbt * (I + H) * sizeof(float);
    if (lds > 96 * 1024) { set_error("%s: NB*(I+H) does not fit the LDS budget", what); return ICAP_ERR_LIMIT; }
    const int unit = w16 ? 8 : 4;
    const int it = ((I > H ? I : H) / unit + 63) / 64;
    if (lstm && w16) cell_dispatch<true, true>(nbt, it, H, lds, st, w_ih, w_hh, b_ih, b_hh, in, NB, h_in, c_in, h_out, c_out, I);
    else if (lstm) cell_dispatch<true, false>(nbt, it, H, lds, st, w_ih, w_hh, b_ih, b_hh, in, NB, h_in, c_in, h_out, c_out, I);
    else if (w16) cell_dispatch<false, true>(nbt, it, H, lds, st, w_ih, w_hh, b_ih, b_hh, in, NB, h_in, c_in, h_out, c_out, I);
    else cell_dispatch<false, false>(nbt, it, H, lds, st, w_ih, w_hh, b_ih, b_hh, in, NB, h_in, c_in, h_out, c_out, I);
    return launch_status(what);
}

int lstm_cell_f32(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* x, int ldx,
                  const float* h_in, const float* c_in, float* h_out, float* c_out, int NB, int I, int H, hipStream_t st) {
    if (!x) { set_error("lstm_cell: bad arguments"); return ICAP_ERR_ARG; }
    return rnn_cell_step_f32(true, w_ih, w_hh, b_ih, b_hh, plain_input(x, ldx), h_in, c_in, h_out, c_out, NB, I, H, st, false);
}

int gru_cell_f32(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* x, int ldx,
                 const float* h_in, float* h_out, int NB, int I, int H, hipStream_t st) {
    if (!x) { set_error("gru_cell: bad arguments"); return ICAP_ERR_ARG; }
    return rnn_cell_step_f32(false, w_ih, w_hh, b_ih, b_hh, plain_input(x, ldx), h_in, nullptr, h_out, nullptr, NB, I, H, st, false);
}

}  // namespace icap
