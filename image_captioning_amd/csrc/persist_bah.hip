// Whole-caption persistent greedy decoder for the Bahdanau LSTMCell decoder with ONE image per attention batch
// (transformer_word_level_bahdnau.py:95-146 as generate_caption of TransformerTest_bahdnau.py:153-191 runs it, argmax form):
// ONE launch runs every step of one or two captions, with the step's weights - the vocabulary projection and W_hh, 18 MB for
// H = 512, V = 6756 - resident in the LDS of a grid of workgroups, one per CU.
//
// The step is that of icap_greedy_bah_fused_f32 (decode_fused.hip): the token-fed input side of the cell is a row of the table
// g_tab[v] = W_ih[:, :E] . emb[v], the context side g_ctx is constant over the caption, and W_hh . h(t) is computed in the same
// pass as the logits of h(t), one step ahead of its use.  So a step is ONE exchange between the workgroups:
//   every workgroup: h(t) = cell(g_tab[tok(t)] + g_ctx + hh(t), c)   (redundantly; h, c, g_ctx live in its LDS)
//                    its rows of fc . h(t)  -> its best (value, index) -> published
//                    its rows of W_hh . h(t) = hh(t + 1)              -> published
//   then polls the G partial arg-max words (-> tok(t + 1)) and reads the 4H recurrent values.
// Words are 64-bit {payload, tag} with device-scope atomic stores / loads as in persist.hip; the buffers are double-buffered
// by step parity (with one exchange per step a fast workgroup may publish step t + 1 while a slow one still reads step t).
// Polling traffic is kept small: a thread polls ONE partial word (producers publish it last); the 4H recurrent words are read
// once after that, each still verified by its tag and re-read if it has not landed.  Every spin is bounded; on a limit the
// workgroup raises `fail`, stops publishing, the grid drains and workgroup 0 fills ids with -1.
#include <mutex>
#include "decode.hpp"

namespace icap {

constexpr int PB_T = 512;                     // threads per workgroup
constexpr int PB_W = PB_T / 64;
constexpr unsigned PB_SPIN_LIMIT = 4000000u;
constexpr int PB_LDS_BUDGET = 160 * 1024 - 1024;

struct PbArgs {
    const float *g_tab, *g_ctx, *w_hh, *fc_w, *fc_scale, *fc_bias;
    const long long* start_ids;
    long long* ids;
    float* logits_out;
    unsigned long long *hbuf, *lbuf;          // [2][NB][4H], [2][G][NB]
    int* fail;
    int H, V, NB, steps, G, r_fc, r_hh;
};

__device__ __forceinline__ void pb_publish(unsigned long long* slot, unsigned lo, unsigned hi) {
    __hip_atomic_store(slot, (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float pb_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float pb_tanh(float x) { return 2.f * pb_sigmoid(2.f * x) - 1.f; }

template <int NBT>
__global__ __launch_bounds__(PB_T) void bah_persist_kernel(const PbArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int H = a.H, V = a.V, G = a.G, g = blockIdx.x, GH = 4 * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wfc = lds;                                   // [r_fc][H]
    float* whh = wfc + a.r_fc * H;                      // [r_hh][H]
    float* fcs = whh + a.r_hh * H;                      // [r_fc] scale, [r_fc] bias (blocks padded to 4 floats)
    float* fcb = fcs + ((a.r_fc + 3) & ~3);
    float* gctx = fcb + ((a.r_fc + 3) & ~3);            // [NB][4H]
    float* cvec = gctx + NBT * GH;                      // [NB][H]
    float* hvec = cvec + NBT * H;                       // [NB][H]
    float* red = hvec + NBT * H;                        // [PB_W][NB][2] arg-max partials of the waves
    __shared__ int s_fail;
    __shared__ int toks[NBT];
    if (tid == 0) s_fail = 0;

    const int row3 = g * a.r_fc, n3 = max(0, min(a.r_fc, V - row3));
    const int row2 = g * a.r_hh, n2 = max(0, min(a.r_hh, GH - row2));
    // ---- weights and the caption's constants into LDS, once
    for (int i = tid * 4; i < n3 * H; i += PB_T * 4) *reinterpret_cast<float4*>(wfc + i) = *reinterpret_cast<const float4*>(a.fc_w + (size_t)row3 * H + i);
    for (int i = tid * 4; i < n2 * H; i += PB_T * 4) *reinterpret_cast<float4*>(whh + i) = *reinterpret_cast<const float4*>(a.w_hh + (size_t)row2 * H + i);
    for (int i = tid; i < n3; i += PB_T) { fcs[i] = a.fc_scale ? a.fc_scale[row3 + i] : 1.f; fcb[i] = a.fc_bias ? a.fc_bias[row3 + i] : 0.f; }
    for (int i = tid; i < NBT * GH; i += PB_T) gctx[i] = a.g_ctx[i];
    // h = c = 0 (init_hidden_state): the first step's recurrent side is W_hh . 0 = 0
    for (int i = tid; i < NBT * H; i += PB_T) cvec[i] = 0.f;
    if (tid < NBT) {
        const long long t0 = a.start_ids[tid];
        toks[tid] = (int)(t0 < 0 ? 0 : (t0 >= V ? V - 1 : t0));
    }
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
                    if (++spins > PB_SPIN_LIMIT) { s_fail = 1; break; }
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
        // ---- h(t) = cell(g_tab[tok] + g_ctx + hh(t), c): two units per thread, redone by every workgroup.  The table row and
        // the eight recurrent words a thread needs are all requested before any is looked at: one round trip, not eight;
        // each recurrent word is verified by its tag and asked for again if it has not landed.
        for (int e2 = tid; e2 < NBT * H2; e2 += PB_T) {
            const int b = e2 / H2, j = (e2 - b * H2) << 1;
            const float* tr = a.g_tab + (size_t)toks[b] * GH + j;
            float2 gt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) gt[q] = *reinterpret_cast<const float2*>(tr + q * H);
            float hh[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) hh[q][0] = hh[q][1] = 0.f;
            if (t > 0) {
                const unsigned tag = (unsigned)t;
                const unsigned long long* hb = a.hbuf + (size_t)((t - 1) & 1) * NBT * GH + (size_t)b * GH + j;
                unsigned long long w[4][2];
                unsigned spins = 0;
                for (;;) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int u = 0; u < 2; ++u) w[q][u] = __hip_atomic_load(hb + q * H + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool all = true;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int u = 0; u < 2; ++u) all = all && (unsigned)(w[q][u] >> 32) == tag;
                    if (all) break;
                    if (++spins > PB_SPIN_LIMIT) { s_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int u = 0; u < 2; ++u) hh[q][u] = __uint_as_float((unsigned)w[q][u]);
            }
            float hv[2], cv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int o = b * GH + j + u;
                const float gi = (u ? gt[0].y : gt[0].x) + gctx[o] + hh[0][u];
                const float gf = (u ? gt[1].y : gt[1].x) + gctx[o + H] + hh[1][u];
                const float gg = (u ? gt[2].y : gt[2].x) + gctx[o + 2 * H] + hh[2][u];
                const float go = (u ? gt[3].y : gt[3].x) + gctx[o + 3 * H] + hh[3][u];
                cv[u] = pb_sigmoid(gf) * cvec[b * H + j + u] + pb_sigmoid(gi) * pb_tanh(gg);
                hv[u] = pb_sigmoid(go) * pb_tanh(cv[u]);
            }
            *reinterpret_cast<float2*>(cvec + b * H + j) = make_float2(cv[0], cv[1]);
            *reinterpret_cast<float2*>(hvec + b * H + j) = make_float2(hv[0], hv[1]);
        }
        __syncthreads();
        if (s_fail) { ok = false; break; }
        // ---- this workgroup's rows: recurrent rows first (published one by one), then the vocabulary rows and their best
        const unsigned ptag = (unsigned)(t + 1);
        unsigned long long* hbw = a.hbuf + (size_t)(t & 1) * NBT * GH;
        float bv[NBT];
        int bi[NBT];
#pragma unroll
        for (int b = 0; b < NBT; ++b) { bv[b] = -INFINITY; bi[b] = 0x1ffff; }
        for (int r = wave; r < n2 + n3; r += PB_W) {
            const bool rec = r < n2;
            const float* wrow = rec ? whh + r * H : wfc + (r - n2) * H;
            float acc[NBT];
#pragma unroll
            for (int b = 0; b < NBT; ++b) acc[b] = 0.f;
            for (int k = lane * 4; k < H; k += 256) {
                const float4 w = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
                for (int b = 0; b < NBT; ++b) acc[b] = dot4(w, *reinterpret_cast<const float4*>(hvec + b * H + k), acc[b]);
            }
#pragma unroll
            for (int b = 0; b < NBT; ++b) acc[b] = wave_sum(acc[b]);
            if (rec) {
                if (lane < NBT) {
                    float v = 0.f;
#pragma unroll
                    for (int b = 0; b < NBT; ++b) v = lane == b ? acc[b] : v;
                    pb_publish(hbw + (size_t)lane * GH + row2 + r, __float_as_uint(v), ptag);
                }
            } else {
                const int n = row3 + r - n2;
#pragma unroll
                for (int b = 0; b < NBT; ++b) {
                    const float v = fmaf(acc[b], fcs[r - n2], fcb[r - n2]);
                    if (a.logits_out && lane == 0) a.logits_out[((size_t)t * NBT + b) * V + n] = v;
                    argmax_combine(bv[b], bi[b], v, n);
                }
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
            for (int w2 = 0; w2 < PB_W; ++w2) argmax_combine(v, ix, red[(w2 * NBT + tid) * 2], __float_as_int(red[(w2 * NBT + tid) * 2 + 1]));
            pb_publish(a.lbuf + ((size_t)(t & 1) * G + g) * NBT + tid, __float_as_uint(v), (ptag << 17) | (unsigned)ix);
        }
        __syncthreads();      // red and hvec are free for the next step
    }
    if (!ok) {
        if (tid == 0) *a.fail = 1;
        if (g == 0) for (int i = tid; i < NBT * a.steps; i += PB_T) a.ids[i] = -1;
    }
}

static int pb_cus() {
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
static int pb_grid() { return std::min(256, pb_cus() / 64 * 64); }

static long long pb_lds_floats(int NB, int H, int r_fc, int r_hh) {
    return (long long)(r_fc + r_hh) * H + 2LL * ((r_fc + 3) & ~3) + (long long)NB * (4LL * H + 2LL * H) + PB_W * NB * 2 + 16;
}

// captions per call the shape allows (0: not supported - use icap_greedy_bah_fused_f32 / icap_greedy_bah_f32)
int greedy_bah_persist_max_nb(int H, int V) {
    const int G = pb_grid();
    if (G < 64 || H <= 0 || V <= 0 || (H % 4) || V > 0x1fff0) return 0;
    const int r_fc = (V + G - 1) / G, r_hh = (4 * H + G - 1) / G;
    for (int nb = PB_T / G; nb >= 1; --nb)
        if (pb_lds_floats(nb, H, r_fc, r_hh) * 4 <= PB_LDS_BUDGET) return nb > 2 ? 2 : nb;
    return 0;
}

long long greedy_bah_persist_ws_bytes(int NB, int H, int V) {
    (void)V;
    return 8LL * (2LL * NB * 4 * H + 2LL * 256 * NB) + 64;
}

template <int NBT>
static int pb_launch(PbArgs& a, size_t lds_bytes, hipStream_t st) {
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] {
        attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&bah_persist_kernel<NBT>), hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_BUDGET);
    });
    if (attr_rc != hipSuccess) { set_error("greedy_bah_persist: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr_rc)); return ICAP_ERR_HIP_BASE + (int)attr_rc; }
    hipLaunchKernelGGL(bah_persist_kernel<NBT>, dim3(a.G), dim3(PB_T), lds_bytes, st, a);
    return launch_status("greedy_bah_persist");
}

int greedy_bah_persist_f32(const float* g_tab, const float* g_ctx, const float* w_hh, const float* fc_w, const float* fc_scale,
                           const float* fc_bias, const long long* start_ids, long long* ids, float* logits_out, void* ws, int NB, int H,
                           int V, int steps, hipStream_t st) {
    if (!g_tab || !g_ctx || !w_hh || !fc_w || !start_ids || !ids || !ws || NB <= 0 || steps <= 0) { set_error("greedy_bah_persist: bad arguments"); return ICAP_ERR_ARG; }
    const int max_nb = greedy_bah_persist_max_nb(H, V);
    if (max_nb == 0) { set_error("greedy_bah_persist: H=%d V=%d does not fit the chip's LDS (or H is not a multiple of 4)", H, V); return ICAP_ERR_LIMIT; }
    if (NB > max_nb) { set_error("greedy_bah_persist: NB=%d exceeds %d captions per call for this shape", NB, max_nb); return ICAP_ERR_LIMIT; }
    if (steps > 10000) { set_error("greedy_bah_persist: at most 10000 steps"); return ICAP_ERR_LIMIT; }
    if (!aligned16(w_hh) || !aligned16(fc_w) || !aligned16(g_tab) || (reinterpret_cast<uintptr_t>(ws) & 7)) {
        set_error("greedy_bah_persist: weights and table must be 16-byte aligned, ws 8-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    PbArgs a;
    a.g_tab = g_tab; a.g_ctx = g_ctx; a.w_hh = w_hh; a.fc_w = fc_w; a.fc_scale = fc_scale; a.fc_bias = fc_bias;
    a.start_ids = start_ids; a.ids = ids; a.logits_out = logits_out;
    a.H = H; a.V = V; a.NB = NB; a.steps = steps;
    a.G = pb_grid();
    a.r_fc = (V + a.G - 1) / a.G;
    a.r_hh = (4 * H + a.G - 1) / a.G;
    a.hbuf = static_cast<unsigned long long*>(ws);
    a.lbuf = a.hbuf + 2LL * NB * 4 * H;
    a.fail = reinterpret_cast<int*>(a.lbuf + 2LL * 256 * NB);
    // tags start at 1 every launch: slots must not carry a previous launch's tags
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)greedy_bah_persist_ws_bytes(NB, H, V), st);
    if (e != hipSuccess) { set_error("greedy_bah_persist: memset: %s", hipGetErrorString(e)); return ICAP_ERR_HIP_BASE + (int)e; }
    const size_t lds_bytes = (size_t)pb_lds_floats(NB, H, a.r_fc, a.r_hh) * 4;
    return NB == 1 ? pb_launch<1>(a, lds_bytes, st) : pb_launch<2>(a, lds_bytes, st);
}

}  // namespace icap
