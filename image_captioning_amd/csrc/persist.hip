// Whole-caption persistent decoder for the transformer decoders (transformer_word_level.py:72-79,
// transformercharmiguel.py:73-80 in greedy form, TransformerTest.py:139-164 with argmax): ONE launch runs all greedy
// steps of up to 8 captions in lock-step, with the step's weights resident on chip.
//
// Why: one caption in flight is a chain of dependent matrix-vector products.  As separate launches every step re-reads
// its weights through L2 / HBM and pays 2-3 kernel boundaries (>= 2-3 us each) - 14 us per step for the char model whose
// step touches only 2.3 MB.  Here the weight rows are split over G workgroups once (char: 32 workgroups x 70 KB of LDS,
// word: 256 x 86 KB), loaded into LDS at launch, and re-used by every step; per step only the activation vectors
// (a few hundred floats) cross between workgroups.
//
// Exchange without barriers.  A step has three dependent phases - h = relu(fc1 x); y = fc2 h + x; logits = out(LN3(y))
// followed by the arg-max - and each phase's consumers need every producer's rows.  Producers publish each value as one
// 64-bit word {value bits, tag} with a device-scope atomic store, tag = 1 + the global phase number; consumers poll the
// very words they need with device-scope atomic loads until the tag matches.  The payload and its "ready" flag arrive
// in the same access, so a phase boundary costs one cross-chip round trip instead of barrier + reload.  A slot is only
// rewritten one step later, which its writer cannot reach before every reader has consumed it (each workgroup's next
// write depends transitively on all workgroups' current reads), so single buffering is safe.
//
// Every polling loop is bounded: on a scheduling surprise (a workgroup not resident) the spin limit trips, the
// workgroup raises the failure flag, stops publishing and leaves; the others time out the same way, the grid drains,
// and workgroup 0 reports the failure by filling `ids` with -1.
#include <mutex>
#include "decode.hpp"
#include "../../include/icap_hip.h"

namespace icap {

constexpr int PT = 512;                      // threads per workgroup (8 waves)
constexpr int PW = PT / 64;
constexpr int P_MAX_NB = 8;
constexpr unsigned P_SPIN_LIMIT = 4000000u;  // ~ seconds; only reached when something is wrong
constexpr int P_LDS_BUDGET = 160 * 1024 - 1024;

struct PersistArgs {
    const float *emb, *ln2_g, *ln2_b, *ln3_g, *ln3_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b, *out_w, *out_b;
    const float* ca;                 // [NB][E] cross-attention rows
    const long long* start_ids;      // [NB]
    long long* ids;                  // [NB][steps]
    float* logits_out;               // [steps][NB][V] or null
    unsigned long long *hbuf, *ybuf, *lbuf;   // exchange slots: [NB][FF], [NB][E], [G][NB]
    int* fail;
    int E, FF, V, NB, steps, G, r1, r2, r3;
};

__device__ __forceinline__ void publish(unsigned long long* slot, float v, unsigned tag) {
    const unsigned long long w = (unsigned long long)__float_as_uint(v) | ((unsigned long long)tag << 32);
    __hip_atomic_store(slot, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// spins until the upper word of the slot, shifted right by `shift`, equals `tag`; returns false when the spin limit trips
__device__ __forceinline__ bool consume(unsigned long long* slot, unsigned tag, int shift, unsigned long long& out) {
    unsigned spins = 0;
    for (;;) {
        const unsigned long long w = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((((unsigned)(w >> 32)) >> shift) == tag) { out = w; return true; }
        if (++spins > P_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// dot products of one LDS-resident weight row with the NB LDS-resident vectors (K % 4 == 0); every lane gets the sums
template <int NBT>
__device__ __forceinline__ void row_dots(const float* __restrict__ wrow, const float* __restrict__ vec, int ldv, int K, int lane, float (&acc)[NBT]) {
#pragma unroll
    for (int b = 0; b < NBT; ++b) acc[b] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 w = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
        for (int b = 0; b < NBT; ++b) acc[b] = dot4(w, *reinterpret_cast<const float4*>(vec + b * ldv + k), acc[b]);
    }
#pragma unroll
    for (int b = 0; b < NBT; ++b) acc[b] = wave_sum(acc[b]);
}

// LayerNorm of row `src` (+ optional `add`) over E elements by one wave, result to dst (LDS)
__device__ __forceinline__ void wave_layernorm(const float* src, const float* add, const float* g, const float* bta, float* dst, int E, int lane) {
    float s = 0.f;
    for (int k = lane; k < E; k += 64) s += src[k] + (add ? add[k] : 0.f);
    const float mean = wave_sum(s) / (float)E;
    float sq = 0.f;
    for (int k = lane; k < E; k += 64) { const float d = src[k] + (add ? add[k] : 0.f) - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) / (float)E + 1e-5f);
    for (int k = lane; k < E; k += 64) dst[k] = (src[k] + (add ? add[k] : 0.f) - mean) * rstd * g[k] + bta[k];
}

template <int NBT>
__global__ __launch_bounds__(PT) void tfm_persist_kernel(const PersistArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int E = a.E, FF = a.FF, V = a.V, G = a.G, g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // LDS carve-up
    float* w1 = lds;                         // [r1][E]
    float* w2 = w1 + a.r1 * E;               // [r2][FF]
    float* w3 = w2 + a.r2 * FF;              // [r3][E]
    float* b1 = w3 + a.r3 * E;               // [r1], each bias block padded to a multiple of 4 floats
    float* b2 = b1 + ((a.r1 + 3) & ~3);      // [r2]
    float* b3 = b2 + ((a.r2 + 3) & ~3);      // [r3]
    float* lnp = b3 + ((a.r3 + 3) & ~3);
    float* ln2g = lnp, *ln2b = lnp + E, *ln3g = lnp + 2 * E, *ln3b = lnp + 3 * E;
    float* x = lnp + 4 * E;                  // [NB][E]  LN2(embedding + cross attention): the step's input
    float* x2 = x + NBT * E;                 // [NB][E]  y, then LN3(y)
    float* h = x2 + NBT * E;                 // [NB][FF]
    float* red = h + NBT * FF;               // [PW][NB][2] per-wave arg-max partials, then [NB] tokens
    __shared__ int s_fail;
    if (tid == 0) s_fail = 0;

    const int row1 = g * a.r1, n1 = max(0, min(a.r1, FF - row1));
    const int row2 = g * a.r2, n2 = max(0, min(a.r2, E - row2));
    const int row3 = g * a.r3, n3 = max(0, min(a.r3, V - row3));
    // ---- weights into LDS, once
    for (int i = tid * 4; i < n1 * E; i += PT * 4) *reinterpret_cast<float4*>(w1 + i) = *reinterpret_cast<const float4*>(a.fc1_w + (size_t)row1 * E + i);
    for (int i = tid * 4; i < n2 * FF; i += PT * 4) *reinterpret_cast<float4*>(w2 + i) = *reinterpret_cast<const float4*>(a.fc2_w + (size_t)row2 * FF + i);
    for (int i = tid * 4; i < n3 * E; i += PT * 4) *reinterpret_cast<float4*>(w3 + i) = *reinterpret_cast<const float4*>(a.out_w + (size_t)row3 * E + i);
    for (int i = tid; i < n1; i += PT) b1[i] = a.fc1_b[row1 + i];
    for (int i = tid; i < n2; i += PT) b2[i] = a.fc2_b[row2 + i];
    for (int i = tid; i < n3; i += PT) b3[i] = a.out_b[row3 + i];
    for (int i = tid; i < E; i += PT) { ln2g[i] = a.ln2_g[i]; ln2b[i] = a.ln2_b[i]; ln3g[i] = a.ln3_g[i]; ln3b[i] = a.ln3_b[i]; }
    // ---- first input: LN2(emb[start] + ca)
    if (wave < NBT) {
        const long long tok = a.start_ids[wave];
        const long long row = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
        for (int k = lane; k < E; k += 64) x2[wave * E + k] = a.emb[(size_t)row * E + k] + a.ca[wave * E + k];
    }
    __syncthreads();
    if (wave < NBT) wave_layernorm(x2 + wave * E, nullptr, ln2g, ln2b, x + wave * E, E, lane);
    __syncthreads();

    bool ok = true;
    for (int t = 0; t < a.steps && ok; ++t) {
        const unsigned tag1 = 3u * t + 1u, tag2 = tag1 + 1u, tag3 = tag1 + 2u;
        // ---- phase 1: h rows = relu(fc1 x + b1)
        for (int r = wave; r < n1; r += PW) {
            float acc[NBT];
            row_dots<NBT>(w1 + r * E, x, E, E, lane, acc);
            if (lane < NBT) {
                float v = 0.f;
#pragma unroll
                for (int b = 0; b < NBT; ++b) v = lane == b ? acc[b] : v;
                publish(a.hbuf + (size_t)lane * FF + row1 + r, fmaxf(v + b1[r], 0.f), tag1);
            }
        }
        // ---- phase 2: gather h, y rows = fc2 h + b2 + x
        // four words per thread are requested before any is looked at (one round trip, not four); a word that has not
        // landed yet is asked for again
        for (int i0 = tid; i0 < NBT * FF; i0 += 4 * PT) {
            unsigned long long w[4];
            unsigned spins = 0;
            bool all;
            do {
                all = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = min(i0 + k * PT, NBT * FF - 1);
                    w[k] = __hip_atomic_load(a.hbuf + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) all = all && (unsigned)(w[k] >> 32) == tag1;
                if (!all) {
                    if (++spins > P_SPIN_LIMIT) { s_fail = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            } while (!all);
            if (!all) break;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * PT;
                if (i < NBT * FF) h[i] = __uint_as_float((unsigned)w[k]);
            }
        }
        __syncthreads();
        if (s_fail) { ok = false; break; }
        for (int r = wave; r < n2; r += PW) {
            float acc[NBT];
            row_dots<NBT>(w2 + r * FF, h, FF, FF, lane, acc);
            if (lane < NBT) {
                float v = 0.f;
#pragma unroll
                for (int b = 0; b < NBT; ++b) v = lane == b ? acc[b] : v;
                publish(a.ybuf + (size_t)lane * E + row2 + r, v + b2[r] + x[lane * E + row2 + r], tag2);
            }
        }
        // ---- phase 3: gather y, x2 = LN3(y), logits rows = out x2 + b3, per-workgroup arg-max
        for (int i = tid; i < NBT * E; i += PT) {
            unsigned long long w;
            if (!consume(a.ybuf + i, tag2, 0, w)) { s_fail = 1; break; }
            x2[i] = __uint_as_float((unsigned)w);
        }
        __syncthreads();
        if (s_fail) { ok = false; break; }
        if (wave < NBT) wave_layernorm(x2 + wave * E, nullptr, ln3g, ln3b, x2 + wave * E, E, lane);
        __syncthreads();
        float bv[NBT];
        int bi[NBT];
#pragma unroll
        for (int b = 0; b < NBT; ++b) { bv[b] = -INFINITY; bi[b] = 0x1ffff; }
        for (int r = wave; r < n3; r += PW) {
            float acc[NBT];
            row_dots<NBT>(w3 + r * E, x2, E, E, lane, acc);
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                const float v = acc[b] + b3[r];
                if (a.logits_out && lane == 0) a.logits_out[((size_t)t * NBT + b) * V + row3 + r] = v;
                argmax_combine(bv[b], bi[b], v, row3 + r);
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
            for (int w = 0; w < PW; ++w) argmax_combine(v, ix, red[(w * NBT + tid) * 2], __float_as_int(red[(w * NBT + tid) * 2 + 1]));
            publish(a.lbuf + (size_t)g * NBT + tid, v, (tag3 << 17) | (unsigned)ix);
        }
        // ---- arg-max over the workgroups' partials (every workgroup, redundantly), token, next input
        if (wave < NBT) {
            float v = -INFINITY;
            int ix = 0x1ffff;
            for (int j = lane; j < G; j += 64) {
                unsigned long long w;
                if (!consume(a.lbuf + (size_t)j * NBT + wave, tag3, 17, w)) { s_fail = 1; break; }
                argmax_combine(v, ix, __uint_as_float((unsigned)w), (int)((unsigned)(w >> 32) & 0x1ffffu));
            }
            wave_argmax(v, ix);
            if (ix >= V) ix = 0;                       // only when every logit is NaN
            if (g == 0 && lane == 0) a.ids[(size_t)wave * a.steps + t] = ix;
            if (t + 1 < a.steps)
                for (int k = lane; k < E; k += 64) x2[wave * E + k] = a.emb[(size_t)ix * E + k] + a.ca[wave * E + k];
        }
        __syncthreads();
        if (s_fail) { ok = false; break; }
        if (t + 1 < a.steps) {
            if (wave < NBT) wave_layernorm(x2 + wave * E, nullptr, ln2g, ln2b, x + wave * E, E, lane);
            __syncthreads();
        }
    }
    if (!ok) {
        if (tid == 0) *a.fail = 1;
        if (g == 0) for (int i = tid; i < NBT * a.steps; i += PT) a.ids[i] = -1;
    }
}

static long long lds_floats(int NB, int E, int FF, int r1, int r2, int r3) {
    return (long long)r1 * E + (long long)r2 * FF + (long long)r3 * E + ((r1 + 3) & ~3) + ((r2 + 3) & ~3) + ((r3 + 3) & ~3) + 4LL * E +
           2LL * NB * E + (long long)NB * FF + PW * NB * 2 + 16;
}

// rows per workgroup for a grid of G workgroups
static void split_rows(int G, int E, int FF, int V, int& r1, int& r2, int& r3) {
    r1 = (FF + G - 1) / G;
    r2 = (E + G - 1) / G;
    r3 = (V + G - 1) / G;
}

// smallest power-of-two grid (>= 8) whose per-workgroup slice fits in LDS for NB captions; 0 if none up to the CU count
static int pick_grid(int NB, int E, int FF, int V, int max_g) {
    for (int G = 8; G <= max_g; G *= 2) {
        int r1, r2, r3;
        split_rows(G, E, FF, V, r1, r2, r3);
        if (lds_floats(NB, E, FF, r1, r2, r3) * 4 <= P_LDS_BUDGET) return G;
    }
    return 0;
}

static int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        (void)hipGetLastError();
    }
    return cus;
}

int greedy_tfm_persist_max_nb(int E, int FF, int V) {
    if (E <= 0 || FF <= 0 || V <= 0 || (E % 4) || (FF % 4) || V > 0x1fff0) return 0;
    for (int nb = P_MAX_NB; nb >= 1; --nb)
        if (pick_grid(nb, E, FF, V, device_cus())) return nb;
    return 0;
}

long long greedy_tfm_persist_ws_bytes(int NB, int E, int FF, int V) {
    (void)V;
    return 8LL * ((long long)NB * FF + (long long)NB * E + 256LL * NB) + 64;
}

template <int NBT>
static int launch_persist(PersistArgs& a, size_t lds_bytes, hipStream_t st) {
    static std::once_flag once;
    static hipError_t attr_rc = hipSuccess;
    std::call_once(once, [] {
        attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&tfm_persist_kernel<NBT>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BUDGET);
    });
    if (attr_rc != hipSuccess) { set_error("greedy_tfm_persist: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr_rc)); return ICAP_ERR_HIP_BASE + (int)attr_rc; }
    hipLaunchKernelGGL(tfm_persist_kernel<NBT>, dim3(a.G), dim3(PT), lds_bytes, st, a);
    return launch_status("greedy_tfm_persist");
}

int greedy_tfm_persist_f32(const icap_tfm_weights* w, const float* ca, const long long* start_ids, long long* ids, float* logits_out,
                           void* ws, int NB, int steps, hipStream_t st) {
    if (!w || !ca || !start_ids || !ids || !ws || NB <= 0 || steps <= 0) { set_error("greedy_tfm_persist: bad arguments"); return ICAP_ERR_ARG; }
    if (!w->emb || !w->ln2_g || !w->ln2_b || !w->ln3_g || !w->ln3_b || !w->fc1_w || !w->fc1_b || !w->fc2_w || !w->fc2_b || !w->out_w || !w->out_b) {
        set_error("greedy_tfm_persist: bad weight descriptor");
        return ICAP_ERR_ARG;
    }
    const int E = w->E, FF = w->FF, V = w->V;
    const int max_nb = greedy_tfm_persist_max_nb(E, FF, V);
    if (max_nb == 0) { set_error("greedy_tfm_persist: E=%d FF=%d V=%d does not fit the chip's LDS (or E, FF not multiples of 4)", E, FF, V); return ICAP_ERR_LIMIT; }
    if (NB > max_nb) { set_error("greedy_tfm_persist: NB=%d exceeds %d captions per call for this shape", NB, max_nb); return ICAP_ERR_LIMIT; }
    if (steps > 10000) { set_error("greedy_tfm_persist: at most 10000 steps"); return ICAP_ERR_LIMIT; }
    if (!aligned16(w->fc1_w) || !aligned16(w->fc2_w) || !aligned16(w->out_w) || (reinterpret_cast<uintptr_t>(ws) & 7)) {
        set_error("greedy_tfm_persist: weights must be 16-byte aligned, ws 8-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    PersistArgs a;
    a.emb = w->emb; a.ln2_g = w->ln2_g; a.ln2_b = w->ln2_b; a.ln3_g = w->ln3_g; a.ln3_b = w->ln3_b;
    a.fc1_w = w->fc1_w; a.fc1_b = w->fc1_b; a.fc2_w = w->fc2_w; a.fc2_b = w->fc2_b; a.out_w = w->out_w; a.out_b = w->out_b;
    a.ca = ca; a.start_ids = start_ids; a.ids = ids; a.logits_out = logits_out;
    a.E = E; a.FF = FF; a.V = V; a.NB = NB; a.steps = steps;
    a.G = pick_grid(NB, E, FF, V, device_cus());
    split_rows(a.G, E, FF, V, a.r1, a.r2, a.r3);
    const long long ws_bytes = greedy_tfm_persist_ws_bytes(NB, E, FF, V);
    a.hbuf = static_cast<unsigned long long*>(ws);
    a.ybuf = a.hbuf + (size_t)NB * FF;
    a.lbuf = a.ybuf + (size_t)NB * E;
    a.fail = reinterpret_cast<int*>(a.lbuf + 256LL * NB);
    // tags start at 1 every launch: slots must not carry a previous launch's tags
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)ws_bytes, st);
    if (e != hipSuccess) { set_error("greedy_tfm_persist: memset: %s", hipGetErrorString(e)); return ICAP_ERR_HIP_BASE + (int)e; }
    const size_t lds_bytes = (size_t)lds_floats(NB, E, FF, a.r1, a.r2, a.r3) * 4;
    switch (NB) {
        case 1: return launch_persist<1>(a, lds_bytes, st);
        case 2: return launch_persist<2>(a, lds_bytes, st);
        case 3: return launch_persist<3>(a, lds_bytes, st);
        case 4: return launch_persist<4>(a, lds_bytes, st);
        case 5: return launch_persist<5>(a, lds_bytes, st);
        case 6: return launch_persist<6>(a, lds_bytes, st);
        case 7: return launch_persist<7>(a, lds_bytes, st);
        default: return launch_persist<8>(a, lds_bytes, st);
    }
}

}  // namespace icap
