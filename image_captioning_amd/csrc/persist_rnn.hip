// Whole-caption persistent greedy decoder for the two-layer LSTM / GRU of DecoderRNN.sample (seq2seqbasic.py:68-91: nn.LSTM or
// nn.GRU with two layers of one width, Linear(H, V), arg-max, embedding of the arg-max as the next input): ONE launch runs every step of
// one or two captions, with the step's matrices - W_ih1, W_hh0, W_hh1 and the vocabulary projection, 26 MB for H = 512,
// V = 6756 - resident in the LDS of a grid of workgroups, one per CU.
//
// The step is that of icap_greedy_rnn2_fused_f32 (decode_fused.hip): the token-fed input side of layer 0 is a row of the table
// g_tab0[v] = W_ih0 . in(v) + b_ih0 + b_hh0 (step 0: g_first, the same for the image feature), and every recurrent product
// W_hh . h(t) is computed a step ahead of its use, in the pass that first holds h(t).  So a step is TWO exchanges:
//   every workgroup: h0(t) = cell(g_tab0[tok(t)] + hh0(t), c0)                       (redundantly; states live in its LDS)
//                    its rows of [W_ih1; W_hh0] . h0(t) = ih1(t), hh0(t + 1)          -> published          (exchange A)
//   every workgroup: h1(t) = cell(ih1(t) + b1 + hh1(t), c1)                           (redundantly)
//                    its rows of W_hh1 . h1(t) = hh1(t + 1) and of fc . h1(t) -> its best (value, index)
//                                                                                     -> published          (exchange B)
//   then polls the G partial arg-max words (-> tok(t + 1)).
// Exchange A is polled on the gate words themselves (eight per thread): a per-workgroup 'rows done' word polled first was
// measured and costs a microsecond per step - the extra barrier and round trip outweigh the lighter polling.
// Words are 64-bit {payload, tag} with device-scope atomic stores / loads as in persist_bah.hip, buffers double-buffered by
// step parity: a workgroup reaches the pass that overwrites a buffer only through an exchange that every other workgroup
// feeds after its last read of that buffer.  Every spin is bounded; on a limit the workgroup raises `fail`, stops
// publishing, the grid drains and workgroup 0 fills ids with -1.
//
// GRU: three gates (r, z, n) instead of four; the n gate keeps its two halves apart (n = tanh(in_n + r * (W_hn h + b_hn))), so the
// recurrent biases b_hh travel separately (hh0_bias, hh1_bias) and count from step 0 on, and there is no cell state.
#include <algorithm>
#include <mutex>
#include "decode.hpp"

namespace icap {

constexpr int PR_T = 512;                     // threads per workgroup
constexpr int PR_W = PR_T / 64;
constexpr unsigned PR_SPIN_LIMIT = 4000000u;
constexpr int PR_LDS_BUDGET = 160 * 1024 - 1024;

struct PrArgs {
    const float *g_tab0, *g_first, *w_hh0, *hh0_bias, *w_ih1, *ih1_bias, *w_hh1, *hh1_bias, *fc_w, *fc_b;
    long long* ids;
    float* logits_out;
    unsigned long long *abuf, *hbuf, *lbuf;   // [2][NB][8H] (ih1 | hh0), [2][NB][4H] (hh1), [2][G][NB]
    int* fail;
    int H, V, NB, steps, G, r_a, r_hh, r_fc;
};

__device__ __forceinline__ void pr_publish(unsigned long long* slot, unsigned lo, unsigned hi) {
    __hip_atomic_store(slot, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float pr_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float pr_tanh(float x) { return 2.f * pr_sigmoid(2.f * x) - 1.f; }

// The gate words of two neighbouring units from a published [NG * H] block (gate q of unit j at q * H + j): all eight are
// requested before any is looked at, each is verified by its tag and the set is asked for again until it is complete.
// Returns false when the spin limit was reached.
template <int NG>
__device__ __forceinline__ bool pr_read_gates(const unsigned long long* base, int H, unsigned tag, float (&out)[NG][2]) {
    unsigned long long w[NG][2];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int q = 0; q < NG; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u) w[q][u] = __hip_atomic_load(base + q * H + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool all = true;
#pragma unroll
        for (int q = 0; q < NG; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u) all = all && (unsigned)(w[q][u] >> 32) == tag;
        if (all) break;
        if (++spins > PR_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int u = 0; u < 2; ++u) out[q][u] = __uint_as_float((unsigned)w[q][u]);
    return true;
}

// one cell update of two neighbouring units: in[q][u] the input side of gate q (biases included), hh[q][u] the recurrent side
// (GRU: with b_hh), state at c (LSTM only) / h
template <bool LSTM, int NG>
__device__ __forceinline__ void pr_cell(const float (&in)[NG][2], const float (&hh)[NG][2], float* c, float* h) {
    float hv[2], cv[2] = {0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if constexpr (LSTM) {
            const float gi = in[0][u] + hh[0][u], gf = in[1][u] + hh[1][u], gg = in[2][u] + hh[2][u], go = in[3][u] + hh[3][u];
            cv[u] = pr_sigmoid(gf) * c[u] + pr_sigmoid(gi) * pr_tanh(gg);
            hv[u] = pr_sigmoid(go) * pr_tanh(cv[u]);
        } else {
            const float r = pr_sigmoid(in[0][u] + hh[0][u]), z = pr_sigmoid(in[1][u] + hh[1][u]);
            const float n = pr_tanh(in[2][u] + r * hh[2][u]);
            hv[u] = (1.f - z) * n + z * h[u];
        }
    }
    if constexpr (LSTM) *reinterpret_cast<float2*>(c) = make_float2(cv[0], cv[1]);
    *reinterpret_cast<float2*>(h) = make_float2(hv[0], hv[1]);
}

template <int NBT, bool LSTM>
__global__ __launch_bounds__(PR_T) void rnn_persist_kernel(const PrArgs a) {
    extern __shared__ __align__(16) float lds[];
    constexpr int NG = LSTM ? 4 : 3;
    const int H = a.H, V = a.V, G = a.G, g = blockIdx.x, GH = NG * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wa = lds;                                    // [r_a][H]   rows of [W_ih1; W_hh0]
    float* whh = wa + a.r_a * H;                        // [r_hh][H]  rows of W_hh1
    float* wfc = whh + a.r_hh * H;                      // [r_fc][H]
    float* fcb = wfc + a.r_fc * H;                      // [r_fc] (padded to 4 floats)
    float* b1 = fcb + ((a.r_fc + 3) & ~3);              // [GH] layer 1's input-side bias (LSTM: b_ih1 + b_hh1)
    float* hb0 = b1 + GH;                               // [GH] b_hh0, [GH] b_hh1 (GRU; zeros for an LSTM)
    float* hb1 = hb0 + GH;
    float* gfirst = hb1 + GH;                           // [NB][GH]
    float* c0 = gfirst + NBT * GH;                      // [NB][H] each
    float* h0 = c0 + NBT * H;
    float* c1 = h0 + NBT * H;
    float* h1 = c1 + NBT * H;
    float* red = h1 + NBT * H;                          // [PR_W][NB][2] arg-max partials of the waves
    __shared__ int s_fail;
    __shared__ int toks[NBT];
    if (tid == 0) s_fail = 0;

    // every matrix is cut the same way: workgroup g owns rows g * r_hh ... of W_ih1, W_hh0 and W_hh1 (r_a = 2 r_hh rows of wa:
    // first the W_ih1 rows, whose products the whole grid waits for, then the W_hh0 rows, needed a step later)
    const int rowh = g * a.r_hh, nh = max(0, min(a.r_hh, GH - rowh));
    const int na = 2 * nh;
    const int rowf = g * a.r_fc, nf = max(0, min(a.r_fc, V - rowf));
    // ---- weights and the caption's constants into LDS, once
    for (int r = 0; r < na; ++r) {
        const float* src = r < nh ? a.w_ih1 + (size_t)(rowh + r) * H : a.w_hh0 + (size_t)(rowh + r - nh) * H;
        for (int i = tid * 4; i < H; i += PR_T * 4) *reinterpret_cast<float4*>(wa + r * H + i) = *reinterpret_cast<const float4*>(src + i);
    }
    for (int i = tid * 4; i < nh * H; i += PR_T * 4) *reinterpret_cast<float4*>(whh + i) = *reinterpret_cast<const float4*>(a.w_hh1 + (size_t)rowh * H + i);
    for (int i = tid * 4; i < nf * H; i += PR_T * 4) *reinterpret_cast<float4*>(wfc + i) = *reinterpret_cast<const float4*>(a.fc_w + (size_t)rowf * H + i);
    for (int i = tid; i < nf; i += PR_T) fcb[i] = a.fc_b ? a.fc_b[rowf + i] : 0.f;
    for (int i = tid; i < GH; i += PR_T) {
        b1[i] = a.ih1_bias ? a.ih1_bias[i] : 0.f;
        hb0[i] = a.hh0_bias ? a.hh0_bias[i] : 0.f;
        hb1[i] = a.hh1_bias ? a.hh1_bias[i] : 0.f;
    }
    for (int i = tid; i < NBT * GH; i += PR_T) gfirst[i] = a.g_first[i];
    // h = c = 0 in both layers: the first step's recurrent sides are W_hh . 0 = 0
    for (int i = tid; i < NBT * H; i += PR_T) { c0[i] = 0.f; c1[i] = 0.f; h0[i] = 0.f; h1[i] = 0.f; }
    if (tid < NBT) toks[tid] = 0;
    __syncthreads();

    const int H2 = H >> 1;
    bool ok = true;
    for (int t = 0; t <= a.steps && ok; ++t) {
        if (t > 0) {
            // ---- tok(t) from the partial arg-max words of step t - 1: thread (b, j) polls workgroup j's word of caption b
            const unsigned tag = (unsigned)t;
            const unsigned long long* lb = a.lbuf + (size_t)((t - 1) & 1) * G * NBT;
            const int b = tid / G, j = tid - b * G;
            float v = -INFINITY;
            int ix = 0x1ffff;
            if (b < NBT) {
                unsigned spins = 0;
                for (;;) {
                    const unsigned long long w = __hip_atomic_load(lb + (size_t)j * NBT + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(w >> 49) == tag) { v = __uint_as_float((unsigned)w); ix = (int)((unsigned)(w >> 32) & 0x1ffffu); break; }
                    if (++spins > PR_SPIN_LIMIT) { s_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            wave_argmax(v, ix);                    // G is a multiple of 64: a wave holds words of one caption
            if (lane == 0) { red[wave * 2] = v; red[wave * 2 + 1] = __int_as_float(ix); }
            __syncthreads();
            if (s_fail) { ok = false; break; }
            if (tid < NBT) {
                const int wpc = G / 64;            // waves per caption
                float bv = -INFINITY;
                int bi = 0x1ffff;
                for (int w2 = 0; w2 < wpc; ++w2) argmax_combine(bv, bi, red[(tid * wpc + w2) * 2], __float_as_int(red[(tid * wpc + w2) * 2 + 1]));
                if (bi >= V) bi = 0;               // only when every logit is NaN
                toks[tid] = bi;
                if (g == 0) a.ids[(size_t)tid * a.steps + (t - 1)] = bi;
            }
            __syncthreads();
            if (t == a.steps) break;
        }
        const unsigned tag_prev = (unsigned)t, tag_now = (unsigned)(t + 1);
        // ---- layer 0: h0(t) = cell(input row + hh0(t), c0), two units per thread, redone by every workgroup
        for (int e2 = tid; e2 < NBT * H2; e2 += PR_T) {
            const int b = e2 / H2, j = (e2 - b * H2) << 1;
            float in[NG][2], hh[NG][2];
            if (t == 0) {
#pragma unroll
                for (int q = 0; q < NG; ++q) { const float2 v = *reinterpret_cast<const float2*>(gfirst + b * GH + q * H + j); in[q][0] = v.x; in[q][1] = v.y; }
            } else {
                const float* tr = a.g_tab0 + (size_t)toks[b] * GH + j;
#pragma unroll
                for (int q = 0; q < NG; ++q) { const float2 v = *reinterpret_cast<const float2*>(tr + q * H); in[q][0] = v.x; in[q][1] = v.y; }
            }
#pragma unroll
            for (int q = 0; q < NG; ++q) hh[q][0] = hh[q][1] = 0.f;
            if (t > 0 && !pr_read_gates<NG>(a.abuf + ((size_t)((t - 1) & 1) * NBT + b) * 2 * GH + GH + j, H, tag_prev, hh)) s_fail = 1;
#pragma unroll
            for (int q = 0; q < NG; ++q) { hh[q][0] += hb0[q * H + j]; hh[q][1] += hb0[q * H + j + 1]; }
            pr_cell<LSTM, NG>(in, hh, c0 + b * H + j, h0 + b * H + j);
        }
        __syncthreads();
        if (s_fail) { ok = false; break; }
        // ---- exchange A: this workgroup's rows of W_ih1 . h0(t) (waited for by everybody: first), then of W_hh0 . h0(t)
        {
            unsigned long long* ab = a.abuf + (size_t)(t & 1) * NBT * 2 * GH;
            for (int r = wave; r < na; r += PR_W) {
                float acc[NBT];
#pragma unroll
                for (int b = 0; b < NBT; ++b) acc[b] = 0.f;
                for (int k = lane * 4; k < H; k += 256) {
                    const float4 w = *reinterpret_cast<const float4*>(wa + r * H + k);
#pragma unroll
                    for (int b = 0; b < NBT; ++b) acc[b] = dot4(w, *reinterpret_cast<const float4*>(h0 + b * H + k), acc[b]);
                }
#pragma unroll
                for (int b = 0; b < NBT; ++b) acc[b] = wave_sum(acc[b]);
                if (lane < NBT) {
                    float v = 0.f;
#pragma unroll
                    for (int b = 0; b < NBT; ++b) v = lane == b ? acc[b] : v;
                    pr_publish(ab + (size_t)lane * 2 * GH + (r < nh ? rowh + r : GH + rowh + r - nh), __float_as_uint(v), tag_now);
                }
            }
        }
        // ---- layer 1: h1(t) = cell(ih1(t) + b1 + hh1(t), c1), redone by every workgroup
        for (int e2 = tid; e2 < NBT * H2; e2 += PR_T) {
            const int b = e2 / H2, j = (e2 - b * H2) << 1;
            float ih[NG][2], hh[NG][2];
#pragma unroll
            for (int q = 0; q < NG; ++q) hh[q][0] = hh[q][1] = 0.f;
            if (t > 0 && !pr_read_gates<NG>(a.hbuf + ((size_t)((t - 1) & 1) * NBT + b) * GH + j, H, tag_prev, hh)) s_fail = 1;
            if (!pr_read_gates<NG>(a.abuf + ((size_t)(t & 1) * NBT + b) * 2 * GH + j, H, tag_now, ih)) s_fail = 1;
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                ih[q][0] += b1[q * H + j]; ih[q][1] += b1[q * H + j + 1];
                hh[q][0] += hb1[q * H + j]; hh[q][1] += hb1[q * H + j + 1];
            }
            pr_cell<LSTM, NG>(ih, hh, c1 + b * H + j, h1 + b * H + j);
        }
        __syncthreads();
        if (s_fail) { ok = false; break; }
        // ---- exchange B: the vocabulary rows and their best first (the next token waits for them), then the recurrent rows of
        // layer 1 (read by the next step's layer-1 cell, an exchange later)
        unsigned long long* hbw = a.hbuf + (size_t)(t & 1) * NBT * GH;
        float bv[NBT];
        int bi[NBT];
#pragma unroll
        for (int b = 0; b < NBT; ++b) { bv[b] = -INFINITY; bi[b] = 0x1ffff; }
        for (int r = wave; r < nf; r += PR_W) {
            const float* wrow = wfc + r * H;
            float acc[NBT];
#pragma unroll
            for (int b = 0; b < NBT; ++b) acc[b] = 0.f;
            for (int k = lane * 4; k < H; k += 256) {
                const float4 w = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
                for (int b = 0; b < NBT; ++b) acc[b] = dot4(w, *reinterpret_cast<const float4*>(h1 + b * H + k), acc[b]);
            }
#pragma unroll
            for (int b = 0; b < NBT; ++b) acc[b] = wave_sum(acc[b]);
            const int n = rowf + r;
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                const float v = acc[b] + fcb[r];
                if (a.logits_out && lane == 0) a.logits_out[((size_t)t * NBT + b) * V + n] = v;
                argmax_combine(bv[b], bi[b], v, n);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < NBT; ++b) { red[(wave * NBT + b) * 2] = bv[b]; red[(wave * NBT + b) * 2 + 1] = __int_as_float(bi[b]); }
        }
        __syncthreads();
        if (tid < NBT) {
            float v = -INFINITY;
            int ix = 0x1ffff;
            for (int w2 = 0; w2 < PR_W; ++w2) argmax_combine(v, ix, red[(w2 * NBT + tid) * 2], __float_as_int(red[(w2 * NBT + tid) * 2 + 1]));
            pr_publish(a.lbuf + ((size_t)(t & 1) * G + g) * NBT + tid, __float_as_uint(v), (tag_now << 17) | (unsigned)ix);
        }
        for (int r = wave; r < nh; r += PR_W) {
            float acc[NBT];
#pragma unroll
            for (int b = 0; b < NBT; ++b) acc[b] = 0.f;
            for (int k = lane * 4; k < H; k += 256) {
                const float4 w = *reinterpret_cast<const float4*>(whh + r * H + k);
#pragma unroll
                for (int b = 0; b < NBT; ++b) acc[b] = dot4(w, *reinterpret_cast<const float4*>(h1 + b * H + k), acc[b]);
            }
#pragma unroll
            for (int b = 0; b < NBT; ++b) acc[b] = wave_sum(acc[b]);
            if (lane < NBT) {
                float v = 0.f;
#pragma unroll
                for (int b = 0; b < NBT; ++b) v = lane == b ? acc[b] : v;
                pr_publish(hbw + (size_t)lane * GH + rowh + r, __float_as_uint(v), tag_now);
            }
        }
        __syncthreads();      // red and the state vectors are free for the next step
    }
    if (!ok) {
        if (tid == 0) *a.fail = 1;
        if (g == 0) for (int i = tid; i < NBT * a.steps; i += PR_T) a.ids[i] = -1;
    }
}

static int pr_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        (void)hipGetLastError();
    }
    return cus;
}

// grid: the largest multiple of 64 workgroups that the CUs hold at one per CU, at most 256 (a wave polls the words of one caption)
static int pr_grid() { return std::min(256, pr_cus() / 64 * 64); }

static long long pr_lds_floats(int NB, int NG, int H, int r_a, int r_hh, int r_fc) {
    return (long long)(r_a + r_hh + r_fc) * H + ((r_fc + 3) & ~3) + 3LL * NG * H + (long long)NB * (NG * H + 4LL * H) + PR_W * NB * 2 + 16;
}

// captions per call the shape allows (0: not supported - use icap_greedy_rnn2_fused_f32 / icap_greedy_rnn_f32)
int greedy_rnn2_persist_max_nb(int is_lstm, int H, int V) {
    const int G = pr_grid(), NG = is_lstm ? 4 : 3;
    if (G < 64 || H <= 0 || V <= 0 || (H % 4) || V > 0x1fff0) return 0;
    const int r_hh = (NG * H + G - 1) / G, r_a = 2 * r_hh, r_fc = (V + G - 1) / G;
    for (int nb = std::min(2, PR_T / G); nb >= 1; --nb)
        if (pr_lds_floats(nb, NG, H, r_a, r_hh, r_fc) * 4 <= PR_LDS_BUDGET) return nb;
    return 0;
}

long long greedy_rnn2_persist_ws_bytes(int NB, int is_lstm, int H, int V) {
    (void)V;
    const long long GH = (is_lstm ? 4LL : 3LL) * H;
    return 8LL * (2LL * NB * 2 * GH + 2LL * NB * GH + 2LL * 256 * NB) + 64;
}

template <int NBT, bool LSTM>
static int pr_launch(PrArgs& a, size_t lds_bytes, hipStream_t st) {
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] {
        attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_persist_kernel<NBT, LSTM>), hipFuncAttributeMaxDynamicSharedMemorySize, PR_LDS_BUDGET);
    });
    if (attr_rc != hipSuccess) { set_error("greedy_rnn2_persist: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr_rc)); return ICAP_ERR_HIP_BASE + (int)attr_rc; }
    hipLaunchKernelGGL((rnn_persist_kernel<NBT, LSTM>), dim3(a.G), dim3(PR_T), lds_bytes, st, a);
    return launch_status("greedy_rnn2_persist");
}

// Operands as icap_greedy_rnn2_fused_f32 takes them.
int greedy_rnn2_persist_f32(int is_lstm, const float* g_tab0, const float* g_first, const float* w_hh0, const float* hh0_bias,
                            const float* w_ih1, const float* ih1_bias, const float* w_hh1, const float* hh1_bias, const float* fc_w,
                            const float* fc_b, long long* ids, float* logits_out, void* ws, int NB, int H, int V, int steps,
                            hipStream_t st) {
    if (!g_tab0 || !g_first || !w_hh0 || !w_ih1 || !w_hh1 || !fc_w || !ids || !ws || NB <= 0 || steps <= 0) {
        set_error("greedy_rnn2_persist: bad arguments");
        return ICAP_ERR_ARG;
    }
    const int max_nb = greedy_rnn2_persist_max_nb(is_lstm, H, V);
    if (max_nb == 0) { set_error("greedy_rnn2_persist: H=%d V=%d does not fit the chip's LDS (or H is not a multiple of 4)", H, V); return ICAP_ERR_LIMIT; }
    if (NB > max_nb) { set_error("greedy_rnn2_persist: NB=%d exceeds %d captions per call for this shape", NB, max_nb); return ICAP_ERR_LIMIT; }
    if (steps > 10000) { set_error("greedy_rnn2_persist: at most 10000 steps"); return ICAP_ERR_LIMIT; }
    if (!aligned16(w_hh0) || !aligned16(w_ih1) || !aligned16(w_hh1) || !aligned16(fc_w) || !aligned16(g_tab0) ||
        (reinterpret_cast<uintptr_t>(g_first) & 7) || (reinterpret_cast<uintptr_t>(ws) & 7)) {
        set_error("greedy_rnn2_persist: weights and table must be 16-byte aligned, g_first and ws 8-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    const int NG = is_lstm ? 4 : 3;
    PrArgs a;
    a.g_tab0 = g_tab0; a.g_first = g_first; a.w_hh0 = w_hh0; a.hh0_bias = hh0_bias; a.w_ih1 = w_ih1; a.ih1_bias = ih1_bias;
    a.w_hh1 = w_hh1; a.hh1_bias = hh1_bias; a.fc_w = fc_w; a.fc_b = fc_b; a.ids = ids; a.logits_out = logits_out;
    a.H = H; a.V = V; a.NB = NB; a.steps = steps;
    a.G = pr_grid();
    a.r_hh = (NG * H + a.G - 1) / a.G;
    a.r_a = 2 * a.r_hh;
    a.r_fc = (V + a.G - 1) / a.G;
    a.abuf = static_cast<unsigned long long*>(ws);
    a.hbuf = a.abuf + 2LL * NB * 2 * NG * H;
    a.lbuf = a.hbuf + 2LL * NB * NG * H;
    a.fail = reinterpret_cast<int*>(a.lbuf + 2LL * 256 * NB);
    // tags start at 1 every launch: slots must not carry a previous launch's tags
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)greedy_rnn2_persist_ws_bytes(NB, is_lstm, H, V), st);
    if (e != hipSuccess) { set_error("greedy_rnn2_persist: memset: %s", hipGetErrorString(e)); return ICAP_ERR_HIP_BASE + (int)e; }
    const size_t lds_bytes = (size_t)pr_lds_floats(NB, NG, H, a.r_a, a.r_hh, a.r_fc) * 4;
    if (is_lstm) return NB == 1 ? pr_launch<1, true>(a, lds_bytes, st) : pr_launch<2, true>(a, lds_bytes, st);
    return NB == 1 ? pr_launch<1, false>(a, lds_bytes, st) : pr_launch<2, false>(a, lds_bytes, st);
}

}  // namespace icap
