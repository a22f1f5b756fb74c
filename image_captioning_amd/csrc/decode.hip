// Weight-streaming kernels for the per-step decoder forward at small batch
// (1..8 captions decoded in lock-step).  At these sizes a step is bound by
// reading every weight once, so each kernel streams fp32 weight rows straight
// into registers with 16-byte loads (one wave per group of output rows), keeps the
// handful of activation vectors in LDS, and fuses what follows the dot products:
// bias / activation, the LSTM or GRU cell update, the vocabulary arg-max.
#include "common.hpp"

namespace icap {

constexpr int GEMV_MAX_NB = 8;

// Sum NB per-lane partials across the wave.  Lane 0..NB-1 style access is not
// needed: every lane ends with all NB totals.
template <int NB>
__device__ __forceinline__ void wave_sum_n(float (&a)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) a[b] = wave_sum(a[b]);
}

// ---------------------------------------------------------------- GEMV (+ optional arg-max partials)
// y[b][n] = act((W[n,:] . x[b,:]) * scale[n] + bias[n]),  b < NB, W row-major [N][K], K % 4 == 0.
// With ARGMAX the workgroup also writes its best (value, index) per b to
// part_val/part_idx[b][blockIdx.x]; y may then be null.
template <int NB, int R, bool ARGMAX>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ W, const float* __restrict__ scale,
                                                   const float* __restrict__ bias, const float* __restrict__ x,
                                                   int ldx, float* __restrict__ y, int ldy, int N, int K, int act,
                                                   float* __restrict__ part_val, int* __restrict__ part_idx) {
    extern __shared__ __attribute__((aligned(16))) float4 xs[];  // [NB][K/4]
    __shared__ float av[4][GEMV_MAX_NB];
    __shared__ int ai[4][GEMV_MAX_NB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int K4 = K >> 2;
    for (int i = tid; i < NB * K4; i += 256) {
        const int b = i / K4, k4 = i - b * K4;
        xs[i] = *reinterpret_cast<const float4*>(x + (size_t)b * ldx + (k4 << 2));
    }
    __syncthreads();

    const int row0 = (blockIdx.x * 4 + wave) * R;
    float acc[R][NB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;

    const float4* W4 = reinterpret_cast<const float4*>(W);
    for (int k4 = lane; k4 < K4; k4 += 64) {
        float4 wv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = row0 + r;
            wv[r] = (n < N) ? W4[(size_t)n * K4 + k4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 xv = xs[b * K4 + k4];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][b] = dot4(wv[r], xv, acc[r][b]);
        }
    }
    float bestv[NB];
    int besti[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { bestv[b] = -INFINITY; besti[b] = 0x7fffffff; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        wave_sum_n<NB>(acc[r]);
        const int n = row0 + r;
        if (n < N) {
            const float sc = scale ? scale[n] : 1.f;
            const float bi = bias ? bias[n] : 0.f;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float v = act_apply(fmaf(acc[r][b], sc, bi), act);
                if (y && lane == b) y[(size_t)b * ldy + n] = v;
                if (ARGMAX) argmax_combine(bestv[b], besti[b], v, n);
            }
        }
    }
    if (ARGMAX) {
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < NB; ++b) { av[wave][b] = bestv[b]; ai[wave][b] = besti[b]; }
        }
        __syncthreads();
        if (tid < NB) {
            float v = av[0][tid];
            int i = ai[0][tid];
            for (int w = 1; w < 4; ++w) argmax_combine(v, i, av[w][tid], ai[w][tid]);
            part_val[(size_t)tid * gridDim.x + blockIdx.x] = v;
            part_idx[(size_t)tid * gridDim.x + blockIdx.x] = i;
        }
    }
}

constexpr int GEMV_R = 4;

int gemv_blocks(int N) { return (N + 4 * GEMV_R - 1) / (4 * GEMV_R); }

template <int NB>
static void gemv_launch(bool argmax, int blocks, size_t lds, hipStream_t st, const float* W, const float* scale,
                        const float* bias, const float* x, int ldx, float* y, int ldy, int N, int K, int act,
                        float* pv, int* pi) {
    if (argmax)
        hipLaunchKernelGGL((gemv_kernel<NB, GEMV_R, true>), dim3(blocks), dim3(256), lds, st, W, scale, bias, x, ldx, y, ldy, N, K, act, pv, pi);
    else
        hipLaunchKernelGGL((gemv_kernel<NB, GEMV_R, false>), dim3(blocks), dim3(256), lds, st, W, scale, bias, x, ldx, y, ldy, N, K, act, pv, pi);
}

int gemv_f32(const float* W, const float* scale, const float* bias, const float* x, int ldx, float* y, int ldy,
             int NB, int N, int K, int act, float* part_val, int* part_idx, hipStream_t st) {
    const bool argmax = part_val != nullptr;
    if (!W || !x || (!y && !argmax) || (argmax && !part_idx) || NB <= 0 || N <= 0 || K <= 0) {
        set_error("gemv: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (NB > GEMV_MAX_NB) { set_error("gemv: batch %d exceeds %d (use the GEMM path)", NB, GEMV_MAX_NB); return ICAP_ERR_LIMIT; }
    if (K % 4 || ldx % 4 || !aligned16(W) || !aligned16(x)) { set_error("gemv: K, ldx %% 4 and 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    const size_t lds = (size_t)NB * K * sizeof(float);
    if (lds > 96 * 1024) { set_error("gemv: NB*K=%d floats do not fit the LDS budget", NB * K); return ICAP_ERR_LIMIT; }
    const int blocks = gemv_blocks(N);
    switch (NB) {
        case 1: gemv_launch<1>(argmax, blocks, lds, st, W, scale, bias, x, ldx, y, ldy, N, K, act, part_val, part_idx); break;
        case 2: gemv_launch<2>(argmax, blocks, lds, st, W, scale, bias, x, ldx, y, ldy, N, K, act, part_val, part_idx); break;
        case 3: gemv_launch<3>(argmax, blocks, lds, st, W, scale, bias, x, ldx, y, ldy, N, K, act, part_val, part_idx); break;
        case 4: gemv_launch<4>(argmax, blocks, lds, st, W, scale, bias, x, ldx, y, ldy, N, K, act, part_val, part_idx); break;
        case 5: gemv_launch<5>(argmax, blocks, lds, st, W, scale, bias, x, ldx, y, ldy, N, K, act, part_val, part_idx); break;
        case 6: gemv_launch<6>(argmax, blocks, lds, st, W, scale, bias, x, ldx, y, ldy, N, K, act, part_val, part_idx); break;
        case 7: gemv_launch<7>(argmax, blocks, lds, st, W, scale, bias, x, ldx, y, ldy, N, K, act, part_val, part_idx); break;
        default: gemv_launch<8>(argmax, blocks, lds, st, W, scale, bias, x, ldx, y, ldy, N, K, act, part_val, part_idx); break;
    }
    return launch_status("gemv");
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

// ---------------------------------------------------------------- fused LSTM cell
// One wave per hidden unit j: the eight weight rows (i,f,g,o of W_ih and W_hh)
// are streamed once and dotted with every caption's x and h; the cell update runs
// in the same wave.  x: [NB][ldx] (first I used), h_in/c_in/h_out/c_out: [NB][H].
template <int NB>
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                        const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                        const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ h_in, const float* __restrict__ c_in,
                                                        float* __restrict__ h_out, float* __restrict__ c_out, int I, int H) {
    extern __shared__ __attribute__((aligned(16))) float4 vs[];  // x: [NB][I/4] then h: [NB][H/4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int I4 = I >> 2, H4 = H >> 2;
    float4* xs = vs;
    float4* hs = vs + NB * I4;
    for (int i = tid; i < NB * I4; i += 256) {
        const int b = i / I4, k4 = i - b * I4;
        xs[i] = *reinterpret_cast<const float4*>(x + (size_t)b * ldx + (k4 << 2));
    }
    for (int i = tid; i < NB * H4; i += 256) hs[i] = reinterpret_cast<const float4*>(h_in)[i];
    __syncthreads();
    const int j = blockIdx.x * 4 + wave;
    if (j >= H) return;
    float acc[4][NB];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[g][b] = 0.f;
    const float4* wi4 = reinterpret_cast<const float4*>(w_ih);
    const float4* wh4 = reinterpret_cast<const float4*>(w_hh);
    for (int k4 = lane; k4 < I4; k4 += 64) {
        float4 wv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) wv[g] = wi4[(size_t)(g * H + j) * I4 + k4];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 xv = xs[b * I4 + k4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g][b] = dot4(wv[g], xv, acc[g][b]);
        }
    }
    for (int k4 = lane; k4 < H4; k4 += 64) {
        float4 wv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) wv[g] = wh4[(size_t)(g * H + j) * H4 + k4];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 hv = hs[b * H4 + k4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g][b] = dot4(wv[g], hv, acc[g][b]);
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) wave_sum_n<NB>(acc[g]);
    float bsum[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bsum[g] = b_ih[g * H + j] + b_hh[g * H + j];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (lane == b) {
            const float ig = sigmoidf_(acc[0][b] + bsum[0]);
            const float fg = sigmoidf_(acc[1][b] + bsum[1]);
            const float gg = tanhf(acc[2][b] + bsum[2]);
            const float og = sigmoidf_(acc[3][b] + bsum[3]);
            const float c = fg * c_in[(size_t)b * H + j] + ig * gg;
            c_out[(size_t)b * H + j] = c;
            h_out[(size_t)b * H + j] = og * tanhf(c);
        }
    }
}

int lstm_cell_f32(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* x,
                  int ldx, const float* h_in, const float* c_in, float* h_out, float* c_out, int NB, int I, int H,
                  hipStream_t st) {
    if (!w_ih || !w_hh || !b_ih || !b_hh || !x || !h_in || !c_in || !h_out || !c_out || NB <= 0 || I <= 0 || H <= 0) {
        set_error("lstm_cell: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (h_in == h_out) { set_error("lstm_cell: h_out must not alias h_in"); return ICAP_ERR_ARG; }
    if (NB > GEMV_MAX_NB) { set_error("lstm_cell: batch %d exceeds %d (use the GEMM path)", NB, GEMV_MAX_NB); return ICAP_ERR_LIMIT; }
    if (I % 4 || H % 4 || ldx % 4 || !aligned16(w_ih) || !aligned16(w_hh) || !aligned16(x) || !aligned16(h_in)) {
        set_error("lstm_cell: I, H, ldx %% 4 and 16-byte alignment required");
        return ICAP_ERR_ALIGN;
    }
    const size_t lds = (size_t)NB * (I + H) * sizeof(float);
    if (lds > 96 * 1024) { set_error("lstm_cell: NB*(I+H) does not fit the LDS budget"); return ICAP_ERR_LIMIT; }
    const dim3 grid((H + 3) / 4), block(256);
#define ICAP_LSTM_CASE(n) case n: hipLaunchKernelGGL(lstm_cell_kernel<n>, grid, block, lds, st, w_ih, w_hh, b_ih, b_hh, x, ldx, h_in, c_in, h_out, c_out, I, H); break;
    switch (NB) {
        ICAP_LSTM_CASE(1) ICAP_LSTM_CASE(2) ICAP_LSTM_CASE(3) ICAP_LSTM_CASE(4)
        ICAP_LSTM_CASE(5) ICAP_LSTM_CASE(6) ICAP_LSTM_CASE(7)
        default: hipLaunchKernelGGL(lstm_cell_kernel<8>, grid, block, lds, st, w_ih, w_hh, b_ih, b_hh, x, ldx, h_in, c_in, h_out, c_out, I, H); break;
    }
#undef ICAP_LSTM_CASE
    return launch_status("lstm_cell");
}

// ---------------------------------------------------------------- fused GRU cell
// Gate order (r, z, n); n = tanh(W_in x + b_in + r * (W_hn h + b_hn)).
template <int NB>
__global__ __launch_bounds__(256) void gru_cell_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                       const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                       const float* __restrict__ x, int ldx,
                                                       const float* __restrict__ h_in, float* __restrict__ h_out, int I, int H) {
    extern __shared__ __attribute__((aligned(16))) float4 vs[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int I4 = I >> 2, H4 = H >> 2;
    float4* xs = vs;
    float4* hs = vs + NB * I4;
    for (int i = tid; i < NB * I4; i += 256) {
        const int b = i / I4, k4 = i - b * I4;
        xs[i] = *reinterpret_cast<const float4*>(x + (size_t)b * ldx + (k4 << 2));
    }
    for (int i = tid; i < NB * H4; i += 256) hs[i] = reinterpret_cast<const float4*>(h_in)[i];
    __syncthreads();
    const int j = blockIdx.x * 4 + wave;
    if (j >= H) return;
    float ai[3][NB], ah[3][NB];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int b = 0; b < NB; ++b) { ai[g][b] = 0.f; ah[g][b] = 0.f; }
    const float4* wi4 = reinterpret_cast<const float4*>(w_ih);
    const float4* wh4 = reinterpret_cast<const float4*>(w_hh);
    for (int k4 = lane; k4 < I4; k4 += 64) {
        float4 wv[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) wv[g] = wi4[(size_t)(g * H + j) * I4 + k4];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 xv = xs[b * I4 + k4];
#pragma unroll
            for (int g = 0; g < 3; ++g) ai[g][b] = dot4(wv[g], xv, ai[g][b]);
        }
    }
    for (int k4 = lane; k4 < H4; k4 += 64) {
        float4 wv[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) wv[g] = wh4[(size_t)(g * H + j) * H4 + k4];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float4 hv = hs[b * H4 + k4];
#pragma unroll
            for (int g = 0; g < 3; ++g) ah[g][b] = dot4(wv[g], hv, ah[g][b]);
        }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) { wave_sum_n<NB>(ai[g]); wave_sum_n<NB>(ah[g]); }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (lane == b) {
            const float r = sigmoidf_((ai[0][b] + b_ih[j]) + (ah[0][b] + b_hh[j]));
            const float z = sigmoidf_((ai[1][b] + b_ih[H + j]) + (ah[1][b] + b_hh[H + j]));
            const float n = tanhf((ai[2][b] + b_ih[2 * H + j]) + r * (ah[2][b] + b_hh[2 * H + j]));
            h_out[(size_t)b * H + j] = (1.f - z) * n + z * h_in[(size_t)b * H + j];
        }
    }
}

int gru_cell_f32(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* x, int ldx,
                 const float* h_in, float* h_out, int NB, int I, int H, hipStream_t st) {
    if (!w_ih || !w_hh || !b_ih || !b_hh || !x || !h_in || !h_out || NB <= 0 || I <= 0 || H <= 0) {
        set_error("gru_cell: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (h_in == h_out) { set_error("gru_cell: h_out must not alias h_in"); return ICAP_ERR_ARG; }
    if (NB > GEMV_MAX_NB) { set_error("gru_cell: batch %d exceeds %d (use the GEMM path)", NB, GEMV_MAX_NB); return ICAP_ERR_LIMIT; }
    if (I % 4 || H % 4 || ldx % 4 || !aligned16(w_ih) || !aligned16(w_hh) || !aligned16(x) || !aligned16(h_in)) {
        set_error("gru_cell: I, H, ldx %% 4 and 16-byte alignment required");
        return ICAP_ERR_ALIGN;
    }
    const size_t lds = (size_t)NB * (I + H) * sizeof(float);
    if (lds > 96 * 1024) { set_error("gru_cell: NB*(I+H) does not fit the LDS budget"); return ICAP_ERR_LIMIT; }
    const dim3 grid((H + 3) / 4), block(256);
#define ICAP_GRU_CASE(n) case n: hipLaunchKernelGGL(gru_cell_kernel<n>, grid, block, lds, st, w_ih, w_hh, b_ih, b_hh, x, ldx, h_in, h_out, I, H); break;
    switch (NB) {
        ICAP_GRU_CASE(1) ICAP_GRU_CASE(2) ICAP_GRU_CASE(3) ICAP_GRU_CASE(4)
        ICAP_GRU_CASE(5) ICAP_GRU_CASE(6) ICAP_GRU_CASE(7)
        default: hipLaunchKernelGGL(gru_cell_kernel<8>, grid, block, lds, st, w_ih, w_hh, b_ih, b_hh, x, ldx, h_in, h_out, I, H); break;
    }
#undef ICAP_GRU_CASE
    return launch_status("gru_cell");
}

}  // namespace icap
