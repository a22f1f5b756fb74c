// Row-wise kernels of the caption decoders (fp32): embedding gather, LayerNorm
// with fused residual, arg-max over the vocabulary, multi-head attention over a
// short key list, Bahdanau additive attention, and the LSTM / GRU gate
// non-linearities that follow a batched gate GEMM.  One wave owns one row (or one
// query/head pair); reductions are 64-lane butterflies, never atomics, so results
// are reproducible run to run.
#include "common.hpp"

namespace icap {

// ---------------------------------------------------------------- embedding
// out[r][:] = table[idx[r]][:] (+ add[r][:] when given)
__global__ void embedding_kernel(const float* __restrict__ table, const long long* __restrict__ idx,
                                 const float* __restrict__ add, float* __restrict__ out, int ldo, int rows,
                                 int E, int V, int* __restrict__ err) {
    const int r = blockIdx.x;
    long long t = idx[r];
    if (t < 0 || t >= V) { if (threadIdx.x == 0 && err) *err = 1; t = 0; }
    const float* src = table + (size_t)t * E;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        float v = src[e];
        if (add) v += add[(size_t)r * E + e];
        out[(size_t)r * ldo + e] = v;
    }
}

int embedding_f32(const float* table, const long long* idx, const float* add, float* out, int ldo, int rows, int E,
                  int V, int* err_flag, hipStream_t st) {
    if (!table || !idx || !out || rows <= 0 || E <= 0 || V <= 0 || ldo < E) { set_error("embedding: bad arguments"); return ICAP_ERR_ARG; }
    hipLaunchKernelGGL(embedding_kernel, dim3(rows), dim3(128), 0, st, table, idx, add, out, ldo, rows, E, V, err_flag);
    return launch_status("embedding");
}

// ---------------------------------------------------------------- LayerNorm
// y[r] = LN(x[r] * pre_scale + res[r % res_rows]) * gamma + beta ; one wave per row.  Rows of up to
// 64 * LN_MAXV elements are read once and kept in registers (mean, then centred variance, then output).
constexpr int LN_MAXV = 16;

typedef float ln_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ln_bf16x2 __attribute__((ext_vector_type(2)));

// bf16 (round to nearest even) of v, as its 16 bits
__device__ __forceinline__ unsigned short ln_bf16_bits(float v) {
    const ln_f32x2 p = {v, 0.f};
    return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(p, ln_bf16x2)) & 0xffffu);
}

// y3 (optional): the result also as its exact three-way bf16 split, planes [3][rows * E] (hi = bf16(v), mid = bf16(v - hi),
// lo = bf16(v - hi - mid)) - what the split-product GEMMs take as a pre-split activation; y may then be null.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        int res_rows, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        int rows, int E, float eps, float pre_scale,
                                                        unsigned short* __restrict__ y3) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    const float* xr = x + (size_t)r * E;
    const float* rr = res ? res + (size_t)(r % res_rows) * E : nullptr;
    float v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int e = lane + 64 * i;
        float t = 0.f;
        if (e < E) {
            t = xr[e] * pre_scale;
            if (rr) t += rr[e];
        }
        v[i] = t;
        s += t;
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int e = lane + 64 * i;
        const float d = (e < E) ? v[i] - mean : 0.f;
        q = fmaf(d, d, q);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int e = lane + 64 * i;
        if (e < E) {
            float o = (v[i] - mean) * rstd * gamma[e] + beta[e];
            const size_t at = (size_t)r * E + e;
            if (y) y[at] = o;
            if (y3) {
                const size_t n = (size_t)rows * E;
                const unsigned short h = ln_bf16_bits(o);
                o -= __uint_as_float((unsigned)h << 16);
                const unsigned short m = ln_bf16_bits(o);
                o -= __uint_as_float((unsigned)m << 16);
                y3[at] = h;
                y3[n + at] = m;
                y3[2 * n + at] = ln_bf16_bits(o);
            }
        }
    }
}

int layernorm_split3_f32(const float* x, const float* res, int res_rows, const float* gamma, const float* beta, float* y,
                         unsigned short* y3, int rows, int E, float eps, float pre_scale, hipStream_t st) {
    if (!x || !gamma || !beta || (!y && !y3) || rows <= 0 || E <= 0 || (res && res_rows <= 0)) {
        set_error("layernorm: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (E > 64 * LN_MAXV) { set_error("layernorm: E=%d exceeds %d", E, 64 * LN_MAXV); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, res, res ? res_rows : 1,
                       gamma, beta, y, rows, E, eps, pre_scale, y3);
    return launch_status("layernorm");
}

int layernorm_f32(const float* x, const float* res, int res_rows, const float* gamma, const float* beta,
                  float* y, int rows, int E, float eps, float pre_scale, hipStream_t st) {
    if (!y) { set_error("layernorm: bad arguments"); return ICAP_ERR_ARG; }
    return layernorm_split3_f32(x, res, res_rows, gamma, beta, y, nullptr, rows, E, eps, pre_scale, st);
}

// ---------------------------------------------------------------- argmax over rows
// idx[r * idx_stride] = first index of the maximum of logits[r][0..V)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ logits, int ld,
                                                          long long* __restrict__ idx, int idx_stride, int V,
                                                          float* __restrict__ top2gap) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float s2[4];
    const int r = blockIdx.x;
    const float* row = logits + (size_t)r * ld;
    float best = -INFINITY, second = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float x = row[v];
        if (x > best) { second = best; best = x; bi = v; }
        else if (x > second) second = x;
    }
    // wave reduce keeping the runner-up
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const float o2 = __shfl_xor(second, o, 64);
        const float lose = (ov > best || (ov == best && oi < bi)) ? best : ov;
        second = fmaxf(fmaxf(second, o2), lose);
        argmax_combine(best, bi, ov, oi);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sv[wave] = best; si[wave] = bi; s2[wave] = second; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = sv[0], sec = s2[0];
        int i = si[0];
        for (int w = 1; w < 4; ++w) {
            const float ov = sv[w];
            const int oi = si[w];
            const float lose = (ov > b || (ov == b && oi < i)) ? b : ov;
            sec = fmaxf(fmaxf(sec, s2[w]), lose);
            argmax_combine(b, i, ov, oi);
        }
        idx[(size_t)r * idx_stride] = i;
        if (top2gap) top2gap[r] = b - sec;
    }
}

int argmax_rows_f32(const float* logits, int ld, long long* idx, int idx_stride, int rows, int V,
                    float* top2gap, hipStream_t st) {
    if (!logits || !idx || rows <= 0 || V <= 0 || ld < V || idx_stride <= 0) { set_error("argmax_rows: bad arguments"); return ICAP_ERR_ARG; }
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, st, logits, ld, idx, idx_stride, V, top2gap);
    return launch_status("argmax_rows");
}

// ---------------------------------------------------------------- top-k sampling head
// Per row of logits [V]: softmax, the k most probable tokens (probabilities and indices, most probable first, lower
// index first on ties) and one token drawn among them with probability proportional to its softmax probability -
// softmax -> topk -> multinomial of the reference's generate_caption - in one workgroup per row.  Every thread keeps its
// strided share of the row in registers (V <= 256 * TOPK_PER_THREAD); a round finds the workgroup's maximum and retires
// it.  The draw uses the counter-based generator: row r of call (seed, offset) consumes uniform01(seed, offset + r).
constexpr int TOPK_PER_THREAD = 32;   // V <= 8192
constexpr int TOPK_MAX_K = 16;

__global__ __launch_bounds__(256) void topk_sample_kernel(const float* __restrict__ logits, int ld, int V, int k,
                                                          unsigned long long seed, unsigned long long offset,
                                                          float* __restrict__ probs, int* __restrict__ idx,
                                                          long long* __restrict__ ids, int ids_stride) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float top_p[TOPK_MAX_K];
    __shared__ int top_i[TOPK_MAX_K];
    const int r = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* row = logits + (size_t)r * ld;
    float v[TOPK_PER_THREAD];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < TOPK_PER_THREAD; ++i) {
        const int c = tid + 256 * i;
        v[i] = c < V ? row[c] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    if (lane == 0) sv[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
    __syncthreads();
    float den = 0.f;
#pragma unroll
    for (int i = 0; i < TOPK_PER_THREAD; ++i) den += (tid + 256 * i < V) ? expf(v[i] - mx) : 0.f;
    den = wave_sum(den);
    if (lane == 0) sv[wave] = den;
    __syncthreads();
    den = ((sv[0] + sv[1]) + sv[2]) + sv[3];
    __syncthreads();
    for (int round = 0; round < k; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < TOPK_PER_THREAD; ++i) argmax_combine(bv, bi, v[i], tid + 256 * i);
        wave_argmax(bv, bi);
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
        __syncthreads();
        bv = sv[0]; bi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) argmax_combine(bv, bi, sv[w], si[w]);
        if (tid == 0) { top_p[round] = expf(bv - mx) / den; top_i[round] = bi; }
        // retire the winner in its owner's registers
#pragma unroll
        for (int i = 0; i < TOPK_PER_THREAD; ++i)
            if (tid + 256 * i == bi) v[i] = -INFINITY;
        __syncthreads();
    }
    if (tid < k) {
        if (probs) probs[(size_t)r * k + tid] = top_p[tid];
        if (idx) idx[(size_t)r * k + tid] = top_i[tid];
    }
    if (tid == 0 && ids) {
        float total = 0.f;
        for (int j = 0; j < k; ++j) total += top_p[j];
        const float u = uniform01(seed, offset + (unsigned long long)r) * total;
        float cum = 0.f;
        int pick = top_i[k - 1];
        for (int j = 0; j < k; ++j) {
            cum += top_p[j];
            if (u < cum) { pick = top_i[j]; break; }
        }
        ids[(size_t)r * ids_stride] = pick;
    }
}

int topk_sample_f32(const float* logits, int ld, int rows, int V, int k, unsigned long long seed, unsigned long long offset,
                    float* probs, int* idx, long long* ids, int ids_stride, hipStream_t st) {
    if (!logits || rows <= 0 || V <= 0 || ld < V || k <= 0 || (ids && ids_stride <= 0) || (!probs && !idx && !ids)) {
        set_error("topk_sample: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (k > TOPK_MAX_K || k > V) { set_error("topk_sample: k=%d exceeds %d (or the vocabulary)", k, TOPK_MAX_K); return ICAP_ERR_LIMIT; }
    if (V > 256 * TOPK_PER_THREAD) { set_error("topk_sample: V=%d exceeds %d", V, 256 * TOPK_PER_THREAD); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(topk_sample_kernel, dim3(rows), dim3(256), 0, st, logits, ld, V, k, seed, offset, probs, idx, ids, ids_stride);
    return launch_status("topk_sample");
}

// ---------------------------------------------------------------- next token (arg-max finalize + gather + LayerNorm)
// One wave per caption: reduce the arg-max partials of the vocabulary GEMM to the token id, store it at
// ids[r * ids_stride], then build the next step's input row out[r] = LN(table[id] + add[r]) (LayerNorm only when
// gamma is given; add optional).  Rows up to 64 * LN_MAXV wide.
__global__ __launch_bounds__(256) void next_token_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx,
                                                         int nparts, long long* __restrict__ ids, int ids_stride,
                                                         const float* __restrict__ table, const float* __restrict__ add,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, float* __restrict__ out, int ldo, int rows, int E) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows) return;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int p = lane; p < nparts; p += 64) argmax_combine(bv, bi, part_val[(size_t)r * nparts + p], part_idx[(size_t)r * nparts + p]);
    wave_argmax(bv, bi);
    if (lane == 0) ids[(size_t)r * ids_stride] = bi;
    if (!table) return;
    const float* src = table + (size_t)bi * E;
    float v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int e = lane + 64 * i;
        float t = 0.f;
        if (e < E) {
            t = src[e];
            if (add) t += add[(size_t)r * E + e];
        }
        v[i] = t;
        s += t;
    }
    float mean = 0.f, rstd = 1.f;
    if (gamma) {
        mean = wave_sum(s) / (float)E;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int e = lane + 64 * i;
            const float d = (e < E) ? v[i] - mean : 0.f;
            q = fmaf(d, d, q);
        }
        rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + eps);
    }
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int e = lane + 64 * i;
        if (e < E) out[(size_t)r * ldo + e] = gamma ? (v[i] - mean) * rstd * gamma[e] + beta[e] : v[i];
    }
}

int next_token_f32(const float* part_val, const int* part_idx, int nparts, long long* ids, int ids_stride,
                   const float* table, const float* add, const float* gamma, const float* beta, float eps, float* out,
                   int ldo, int rows, int E, hipStream_t st) {
    if (!part_val || !part_idx || !ids || nparts <= 0 || rows <= 0 || ids_stride <= 0 ||
        (table && (!out || E <= 0 || ldo < E)) || (gamma && !beta)) {
        set_error("next_token: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (table && E > 64 * LN_MAXV) { set_error("next_token: E=%d exceeds %d", E, 64 * LN_MAXV); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(next_token_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, part_val, part_idx, nparts, ids, ids_stride,
                       table, add, gamma, beta, eps, out, ldo, rows, E);
    return launch_status("next_token");
}

// ---------------------------------------------------------------- greedy decode as a walk over a next-token table
// The transformer decoders of the reference have no self-attention and no positional encoding in use
// (transformer_word_level.py:72-79): the logits after a token depend on that token and the image only, so greedy
// decoding is the iteration t -> next[t] of a per-image table.  next: [V][B] (row v*B + b: the arg-max after token v
// for image b), start [B], ids [B][steps].  One workgroup per image: its table column goes to LDS, one lane walks it,
// then every lane copies the logits rows of the visited tokens (logits_tab [V*B][ldl] -> logits_out [steps][B][V]).
__global__ __launch_bounds__(256) void table_walk_kernel(const long long* __restrict__ next, const long long* __restrict__ start,
                                                         long long* __restrict__ ids, const float* __restrict__ logits_tab, int ldl,
                                                         float* __restrict__ logits_out, int V, int B, int steps) {
    extern __shared__ int walk_lds[];     // [V] next token of this image, then [steps] the tokens fed at each step
    int* col = walk_lds;
    int* fed = walk_lds + V;
    const int b = blockIdx.x;
    for (int v = threadIdx.x; v < V; v += 256) {
        const long long n = next[(size_t)v * B + b];
        col[v] = n < 0 ? 0 : (n >= V ? V - 1 : (int)n);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long s = start[b];
        int tok = s < 0 ? 0 : (s >= V ? V - 1 : (int)s);
        for (int t = 0; t < steps; ++t) {
            fed[t] = tok;
            tok = col[tok];
            ids[(size_t)b * steps + t] = tok;
        }
    }
    if (!logits_out) return;
    __syncthreads();
    for (int i = threadIdx.x; i < steps * V; i += 256) {
        const int t = i / V, v = i - t * V;
        logits_out[((size_t)t * B + b) * V + v] = logits_tab[((size_t)fed[t] * B + b) * ldl + v];
    }
}

int table_walk(const long long* next, const long long* start, long long* ids, const float* logits_tab, int ldl, float* logits_out,
               int V, int B, int steps, hipStream_t st) {
    if (!next || !start || !ids || V <= 0 || B <= 0 || steps <= 0 || (logits_out && (!logits_tab || ldl < V))) {
        set_error("table_walk: bad arguments");
        return ICAP_ERR_ARG;
    }
    if ((long long)(V + steps) * 4 > 60 * 1024) { set_error("table_walk: V + steps = %d exceeds the 15360 entries that fit in LDS", V + steps); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(table_walk_kernel, dim3(B), dim3(256), (size_t)(V + steps) * 4, st, next, start, ids, logits_tab, ldl, logits_out, V, B, steps);
    return launch_status("table_walk");
}

// ---------------------------------------------------------------- multi-head attention core
// q: [Lq][N][E]  k, v: [Lk][Nk][E] with Nk == N or Nk == 1 (keys shared by every
// batch column).  out[lq][n][h*hd + d] = sum_s softmax_s(q.k_s / sqrt(hd)) v_s.
// One wave per (lq, n, head); the softmax runs online over the keys.
__global__ __launch_bounds__(256) void mha_core_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ out,
                                                       int Lq, int Lk, int N, int Nk, int E, int heads) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + wave;  // over Lq*N*heads
    if (item >= (long long)Lq * N * heads) return;
    const int h = (int)(item % heads);
    const long long ln = item / heads;
    const int n = (int)(ln % N);
    const int hd = E / heads;
    const float scaling = 1.0f / sqrtf((float)hd);
    const float* qp = q + (size_t)ln * E + (size_t)h * hd;
    const int nk = (Nk == 1) ? 0 : n;
    constexpr int MAXD = 8;  // head_dim <= 512
    float qr[MAXD], acc[MAXD];
#pragma unroll
    for (int i = 0; i < MAXD; ++i) {
        const int d = lane + 64 * i;
        qr[i] = (d < hd) ? qp[d] * scaling : 0.f;
        acc[i] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int s = 0; s < Lk; ++s) {
        const float* kp = k + ((size_t)s * Nk + nk) * E + (size_t)h * hd;
        const float* vp = v + ((size_t)s * Nk + nk) * E + (size_t)h * hd;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < MAXD; ++i) {
            const int d = lane + 64 * i;
            if (d < hd) dot = fmaf(qr[i], kp[d], dot);
        }
        dot = wave_sum(dot);
        const float mn = fmaxf(m, dot);
        const float corr = expf(m - mn);  // exp(-inf) = 0 on the first key
        const float p = expf(dot - mn);
        l = l * corr + p;
#pragma unroll
        for (int i = 0; i < MAXD; ++i) {
            const int d = lane + 64 * i;
            if (d < hd) acc[i] = fmaf(p, vp[d], acc[i] * corr);
        }
        m = mn;
    }
    const float inv = 1.0f / l;
    float* op = out + (size_t)ln * E + (size_t)h * hd;
#pragma unroll
    for (int i = 0; i < MAXD; ++i) {
        const int d = lane + 64 * i;
        if (d < hd) op[d] = acc[i] * inv;
    }
}

int mha_core_f32(const float* q, const float* k, const float* v, float* out, int Lq, int Lk, int N, int Nk,
                 int E, int heads, hipStream_t st) {
    if (!q || !k || !v || !out || Lq <= 0 || Lk <= 0 || N <= 0 || E <= 0 || heads <= 0 || E % heads ||
        (Nk != N && Nk != 1)) {
        set_error("mha_core: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (E / heads > 512) { set_error("mha_core: head_dim %d exceeds 512", E / heads); return ICAP_ERR_LIMIT; }
    const long long items = (long long)Lq * N * heads;
    const long long blocks = (items + 3) / 4;
    if (blocks > 0x7fffffffLL) { set_error("mha_core: too large"); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(mha_core_kernel, dim3((unsigned)blocks), dim3(256), 0, st, q, k, v, out, Lq, Lk, N, Nk, E, heads);
    return launch_status("mha_core");
}

// ---------------------------------------------------------------- Bahdanau additive attention
// score[i][j] = a_b + sum_a a_w[a] * tanh(u[j][a] + w[i][a]);  alpha[i][:] = softmax_j(score[i][:]);
// context[i][:] = sum_j alpha[i][j] * feat[j][:].   One workgroup of 16 waves per query i: a wave scores four keys at a
// time (lanes over a, independent accumulators so the loads of all four are in flight together); the context sum over j
// is split four ways across the workgroup and combined through LDS in a fixed order.
constexpr int BAH_MAX_B = 4096;
constexpr int BAH_THREADS = 1024;

__global__ __launch_bounds__(BAH_THREADS) void bahdanau_attention_kernel(
    const float* __restrict__ u, const float* __restrict__ w, const float* __restrict__ a_w, float a_b,
    const float* __restrict__ feat, float* __restrict__ alpha, float* __restrict__ context, int B, int A, int E) {
    __shared__ float sc[BAH_MAX_B];
    __shared__ float redv[16];
    __shared__ float part[4][256];
    const int i = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* wi = w + (size_t)i * A;
    for (int j0 = wave; j0 < B; j0 += 64) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int a = lane; a < A; a += 64) {
            const float wa = wi[a], aw = a_w[a];
            float uv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = j0 + 16 * q;
                uv[q] = u[(size_t)(j < B ? j : j0) * A + a];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] = fmaf(aw, tanh_fast(uv[q] + wa), s[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = wave_sum(s[q]);
            if (lane == 0 && j0 + 16 * q < B) sc[j0 + 16 * q] = t + a_b;
        }
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = tid; j < B; j += BAH_THREADS) m = fmaxf(m, sc[j]);
    m = wave_max(m);
    if (lane == 0) redv[wave] = m;
    __syncthreads();
    m = redv[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) m = fmaxf(m, redv[k]);
    __syncthreads();
    float l = 0.f;
    for (int j = tid; j < B; j += BAH_THREADS) {
        const float p = expf(sc[j] - m);
        sc[j] = p;
        l += p;
    }
    l = wave_sum(l);
    if (lane == 0) redv[wave] = l;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += redv[k];
    const float inv = 1.0f / tot;
    for (int j = tid; j < B; j += BAH_THREADS) {
        const float p = sc[j] * inv;
        sc[j] = p;
        if (alpha) alpha[(size_t)i * B + j] = p;
    }
    __syncthreads();
    // context: thread (q, t) sums keys [q * Bq, (q + 1) * Bq) for column e = e0 + t
    const int q = tid >> 8, t = tid & 255;
    const int Bq = (B + 3) / 4, jlo = q * Bq, jhi = (jlo + Bq < B) ? jlo + Bq : B;
    for (int e0 = 0; e0 < E; e0 += 256) {
        const int e = e0 + t;
        float c = 0.f;
        if (e < E) {
#pragma unroll 8
            for (int j = jlo; j < jhi; ++j) c = fmaf(sc[j], feat[(size_t)j * E + e], c);
        }
        part[q][t] = c;
        __syncthreads();
        if (q == 0 && e < E) context[(size_t)i * E + e] = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
        __syncthreads();
    }
}

int bahdanau_attention_f32(const float* u, const float* w, const float* a_w, float a_b, const float* feat,
                           float* alpha, float* context, int B, int A, int E, hipStream_t st) {
    if (!u || !w || !a_w || !feat || !context || B <= 0 || A <= 0 || E <= 0) { set_error("bahdanau_attention: bad arguments"); return ICAP_ERR_ARG; }
    if (B > BAH_MAX_B) { set_error("bahdanau_attention: B=%d exceeds %d", B, BAH_MAX_B); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(bahdanau_attention_kernel, dim3(B), dim3(BAH_THREADS), 0, st, u, w, a_w, a_b, feat, alpha, context, B, A, E);
    return launch_status("bahdanau_attention");
}

// ---------------------------------------------------------------- recurrent gate non-linearities
// LSTM: gates[b][4H] = (i, f, g, o) pre-activations with every bias already added.
__global__ void lstm_pointwise_kernel(const float* __restrict__ gates, const float* __restrict__ c_in,
                                      float* __restrict__ h_out, float* __restrict__ c_out, int H, long long total) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const long long b = t / H;
    const int j = (int)(t - b * H);
    const float* g = gates + (size_t)b * 4 * H;
    const float ig = sigmoidf_(g[j]);
    const float fg = sigmoidf_(g[H + j]);
    const float gg = tanhf(g[2 * H + j]);
    const float og = sigmoidf_(g[3 * H + j]);
    const float c = fg * c_in[t] + ig * gg;
    c_out[t] = c;
    h_out[t] = og * tanhf(c);
}

int lstm_pointwise_f32(const float* gates, const float* c_in, float* h_out, float* c_out, int B, int H, hipStream_t st) {
    if (!gates || !c_in || !h_out || !c_out || B <= 0 || H <= 0) { set_error("lstm_pointwise: bad arguments"); return ICAP_ERR_ARG; }
    const long long total = (long long)B * H;
    hipLaunchKernelGGL(lstm_pointwise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gates, c_in, h_out, c_out, H, total);
    return launch_status("lstm_pointwise");
}

// GRU: gi = W_ih x + b_ih, gh = W_hh h + b_hh, both [b][3H] in (r, z, n) order.
__global__ void gru_pointwise_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                     const float* __restrict__ h_in, float* __restrict__ h_out, int H, long long total) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const long long b = t / H;
    const int j = (int)(t - b * H);
    const float* a = gi + (size_t)b * 3 * H;
    const float* c = gh + (size_t)b * 3 * H;
    const float r = sigmoidf_(a[j] + c[j]);
    const float z = sigmoidf_(a[H + j] + c[H + j]);
    const float n = tanhf(a[2 * H + j] + r * c[2 * H + j]);
    h_out[t] = (1.f - z) * n + z * h_in[t];
}

int gru_pointwise_f32(const float* gi, const float* gh, const float* h_in, float* h_out, int B, int H, hipStream_t st) {
    if (!gi || !gh || !h_in || !h_out || B <= 0 || H <= 0) { set_error("gru_pointwise: bad arguments"); return ICAP_ERR_ARG; }
    const long long total = (long long)B * H;
    hipLaunchKernelGGL(gru_pointwise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, gi, gh, h_in, h_out, H, total);
    return launch_status("gru_pointwise");
}

}  // namespace icap
