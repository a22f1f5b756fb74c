// Single-caption greedy decoding of the recurrent decoders with FEWER DEPENDENT LAUNCHES per step.
//
// At one caption in flight a weight-streaming launch costs about 5 us whatever it streams (launch, two or three dependent
// memory round trips, a reduction), so a step's time is its number of dependent launches.  Two observations cut that number:
//   * everything that depends on the new token only through the INPUT side of a cell is a table look-up: W_ih . in(token)
//     is one row of a [V][G H] table made once per weight version;
//   * the recurrent side W_hh . h(t) depends on the state alone, so it can be computed by the launch that projects h(t) on
//     the vocabulary, as extra row segments of the same weight-streaming kernel, a step AHEAD of its use.
// A cell is then a gather and a pointwise update - cheap enough that EVERY workgroup of the next weight-streaming launch
// redoes it in its prologue instead of a launch of its own.  Result: the Bahdanau LSTMCell decoder takes ONE launch per
// step (vocabulary projection + W_hh rows; icap_greedy_bah_fused_f32) and the two-layer LSTM / GRU of the seq2seq model TWO
// (layer 1's input rows | vocabulary projection + both layers' recurrent rows; icap_greedy_rnn2_fused_f32), against two
// and three with one launch per cell.  The prologue is redone by every workgroup (hundreds), so these drivers are meant
// for the one- or two-caption latency case; larger groups stay on the per-cell drivers.
#include "decode.hpp"

namespace icap {

#define FZ_TRY(expr)              \
    do {                          \
        const int rc_ = (expr);   \
        if (rc_ != ICAP_OK) return rc_; \
    } while (0)

__device__ __forceinline__ float4 f4_zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4_sel(bool ok, float4 a) { return ok ? a : f4_zero(); }

constexpr int FZ_MAX_NB = 8;
constexpr int FZ_ROWS = 16;          // weight rows per workgroup (four waves x four rows)
constexpr int FZ_PART = 4096;        // arg-max partials (all captions) staged in LDS

struct FzSeg {                       // y[b][n] = (W[n, :] . x_sel[b, :]) * scale[n] + bias[n], n < N
    const float* W = nullptr;
    const float* scale = nullptr;
    const float* bias = nullptr;
    float* y = nullptr;
    int ldy = 0, N = 0, xsel = 0, amax = 0;
};

struct FzArgs {
    FzSeg seg[3];
    int nseg = 0, blk0[4] = {0, 0, 0, 0};     // first workgroup of each segment
    int H = 0, G = 4;                           // state width, gates per cell (4: LSTM, 3: GRU)
    // prologue: h = cell(tab[token] + add_a | add_b, state) -> x_0 (LDS); zero_h: x_0 = 0 instead (recurrent rows of a zero state)
    int zero_h = 0;
    const float* tab = nullptr;                 // [V][G H] input side per token, or null
    int V = 0;
    const float* add_a = nullptr;               // [NB][G H] input side (added to the table row), or null
    const float* add_b = nullptr;               // [NB][G H] recurrent side W_hh h(t-1) (+ b_hh), or null (= 0)
    const float* c_in = nullptr;                // LSTM cell state [NB][H]
    const float* h_prev = nullptr;              // GRU previous state [NB][H]
    float* c_out = nullptr;
    float* h_out = nullptr;                     // written by workgroup 0
    const float* x1 = nullptr;                  // second input vector [NB][H] (plain), or null
    // token: arg-max of the previous projection's partials, or tok_in, or none (tab == null)
    const float* pv_in = nullptr;
    const int* pi_in = nullptr;
    int nparts_in = 0;
    const long long* tok_in = nullptr;
    long long* ids_out = nullptr;
    int ids_stride = 0;
    // arg-max partials of the segment with amax set: [NB][blocks of that segment]
    float* pv_out = nullptr;
    int* pi_out = nullptr;
};

// sigmoid / tanh on the hardware exponential and reciprocal (one ulp each): the cell update sits on the step's critical
// path in every workgroup, and the library functions cost several times the instructions for accuracy far inside the
// decoders' tolerance (absolute error ~1e-7 on values in [-1, 1]).
__device__ __forceinline__ float fz_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * x)); }
__device__ __forceinline__ float fz_tanh(float x) { return 2.f * fz_sigmoid(2.f * x) - 1.f; }

constexpr int FZ_PW = 8;             // arg-max partials per lane and caption on the register path (vocabulary blocks <= 512)

template <int NB>
__global__ __launch_bounds__(256) void fused_rows_kernel(const FzArgs a, int nb) {
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [2][NB][H]
    __shared__ float red[FZ_ROWS][NB];
    __shared__ float spv[FZ_PART];
    __shared__ int spi[FZ_PART];
    __shared__ int toks[FZ_MAX_NB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int H = a.H, H4 = H >> 2, H2 = H >> 1, GH = a.G * H;
    int s = 0;
    if (a.nseg > 1 && (int)blockIdx.x >= a.blk0[1]) s = 1;
    if (a.nseg > 2 && (int)blockIdx.x >= a.blk0[2]) s = 2;
    const FzSeg sg = a.seg[s];
    const int lblk = blockIdx.x - a.blk0[s];
    const int row0 = lblk * FZ_ROWS + wave * 4;

    // Loads are issued in the order their values are needed, everything before anything is waited for (they return in
    // order): the arg-max partials, the token-independent operands of this thread's first cell element, then the weight
    // rows.  All unconditional, from clamped or stand-in addresses, and masked afterwards - a conditional load makes the
    // compiler wait at every one, and a dozen serialised round trips would be the whole cost of a step.
    const bool use_tok = a.tab != nullptr;
    const bool lstm = a.G == 4;
    const int n2 = nb * H2;                               // cell elements (two units each) of the live captions
    const int ptotal = nb * a.nparts_in;
    const int pw = (a.nparts_in + 63) >> 6;
    const bool part_regs = NB <= 2 && pw <= FZ_PW;       // each wave reduces the partials in its own registers: no LDS, no barrier
    constexpr int PU = NB <= 2 ? NB * FZ_PW : FZ_PART / 256;
    float plv[PU];
    int pli[PU];
    if (use_tok && a.pv_in) {
        if (part_regs) {
#pragma unroll
            for (int b = 0; b < (NB <= 2 ? NB : 0); ++b)
#pragma unroll
                for (int u = 0; u < FZ_PW; ++u) {
                    if (u < pw && b < nb) {
                        const int e = b * a.nparts_in + min(u * 64 + lane, a.nparts_in - 1);
                        plv[b * FZ_PW + u] = a.pv_in[e];
                        pli[b * FZ_PW + u] = a.pi_in[e];
                    }
                }
        } else {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if (u * 256 < ptotal) {
                    const int e = min(tid + 256 * u, ptotal - 1);
                    plv[u] = a.pv_in[e];
                    pli[u] = a.pi_in[e];
                }
            }
        }
    }
    struct CellRegs { float2 ia[4], hb[4], st, x1; };
    auto cell_issue = [&](int e2, CellRegs& r) {
        const int ec = min(e2, max(n2 - 1, 0));
        const int b = ec / H2, j2 = ec - b * H2;
        const float* pb = a.add_b + (size_t)b * GH + (j2 << 1);
        const float* pa = a.add_a ? a.add_a + (size_t)b * GH + (j2 << 1) : pb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t o = (size_t)min(q, a.G - 1) * H;
            r.ia[q] = *reinterpret_cast<const float2*>(pa + o);
            r.hb[q] = *reinterpret_cast<const float2*>(pb + o);
        }
        const float* sp = lstm ? a.c_in : a.h_prev;
        r.st = *reinterpret_cast<const float2*>(sp + (size_t)b * H + (j2 << 1));
        const float* xp = a.x1 ? a.x1 : sp;
        r.x1 = *reinterpret_cast<const float2*>(xp + (size_t)b * H + (j2 << 1));
    };
    CellRegs cr;
    if (!a.zero_h) cell_issue(tid, cr);

    // ---- this wave's four weight rows
    constexpr int IT = 2;
    const float4* W4 = reinterpret_cast<const float4*>(sg.W);
    float4 wv[4][IT];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4* row = W4 + (size_t)min(row0 + r, sg.N - 1) * H4;
#pragma unroll
        for (int i = 0; i < IT; ++i) wv[r][i] = row[min(lane + 64 * i, H4 - 1)];
    }
    // epilogue operands of the threads that finish a (row, caption) pair
    const int er = tid / NB, eb = tid - er * NB;
    const int en = lblk * FZ_ROWS + er;
    const bool e_ok = tid < FZ_ROWS * NB && en < sg.N && eb < nb;
    const float e_scale = (e_ok && sg.scale) ? sg.scale[en] : 1.f;
    const float e_bias = (e_ok && sg.bias) ? sg.bias[en] : 0.f;

    // ---- token per caption
    int tokr[NB <= 2 ? NB : 1];
#pragma unroll
    for (int b = 0; b < (NB <= 2 ? NB : 1); ++b) tokr[b] = 0;
    if (use_tok) {
        if (a.pv_in && part_regs) {
#pragma unroll
            for (int b = 0; b < (NB <= 2 ? NB : 0); ++b) {
                float v = -INFINITY;
                int i = 0x7fffffff;
#pragma unroll
                for (int u = 0; u < FZ_PW; ++u)
                    if (u < pw && b < nb && u * 64 + lane < a.nparts_in) argmax_combine(v, i, plv[b * FZ_PW + u], pli[b * FZ_PW + u]);
                wave_argmax(v, i);
                tokr[b] = (i == 0x7fffffff) ? 0 : i;                     // all-NaN row
            }
        } else {
            if (a.pv_in) {
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    const int e = tid + 256 * u;
                    if (e < ptotal) { spv[e] = plv[u]; spi[e] = pli[u]; }
                }
                __syncthreads();
                for (int b = wave; b < nb; b += 4) {
                    float v = -INFINITY;
                    int i = 0x7fffffff;
                    for (int p = lane; p < a.nparts_in; p += 64) argmax_combine(v, i, spv[b * a.nparts_in + p], spi[b * a.nparts_in + p]);
                    wave_argmax(v, i);
                    if (lane == 0) toks[b] = (i == 0x7fffffff) ? 0 : i;
                }
            } else if (tid < nb) {
                toks[tid] = (int)a.tok_in[tid];
            }
            __syncthreads();
        }
    }
    auto token_of = [&](int b) -> int {
        if (!(a.pv_in && part_regs)) return toks[b];
        int t = tokr[0];
#pragma unroll
        for (int k = 1; k < (NB <= 2 ? NB : 1); ++k) t = b == k ? tokr[k] : t;
        return t;
    };
    if (use_tok && blockIdx.x == 0 && a.ids_out && tid < nb) a.ids_out[(size_t)tid * a.ids_stride] = token_of(tid);

    // ---- x_0 = h(t): the cell update, redone by every workgroup (a table row and a few transcendental functions per unit)
    auto cell_finish = [&](int e2, const CellRegs& r) {
        if (e2 >= NB * H2) return;
        float2 hv = make_float2(0.f, 0.f), xv1 = make_float2(0.f, 0.f);
        if (e2 < n2) {
            const int b = e2 / H2, j2 = e2 - b * H2;
            float2 tb[4];
            const int tok = use_tok ? min(max(token_of(b), 0), a.V - 1) : 0;
            const float* pt = use_tok ? a.tab + (size_t)tok * GH + (j2 << 1) : a.add_b + (size_t)b * GH + (j2 << 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) tb[q] = *reinterpret_cast<const float2*>(pt + (size_t)min(q, a.G - 1) * H);
            const float mt = use_tok ? 1.f : 0.f, ma = a.add_a ? 1.f : 0.f;
            float ih[4][2], hh[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ih[q][0] = mt * tb[q].x + ma * r.ia[q].x; ih[q][1] = mt * tb[q].y + ma * r.ia[q].y;
                hh[q][0] = r.hb[q].x; hh[q][1] = r.hb[q].y;
            }
            const float sv[2] = {r.st.x, r.st.y};
            float h[2], c[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (lstm) {          // i, f, g, o
                    const float ig = fz_sigmoid(ih[0][u] + hh[0][u]), fg = fz_sigmoid(ih[1][u] + hh[1][u]);
                    const float gg = fz_tanh(ih[2][u] + hh[2][u]), og = fz_sigmoid(ih[3][u] + hh[3][u]);
                    c[u] = fg * sv[u] + ig * gg;
                    h[u] = og * fz_tanh(c[u]);
                } else {             // r, z, n with the recurrent half of n gated by r
                    const float rg = fz_sigmoid(ih[0][u] + hh[0][u]), zg = fz_sigmoid(ih[1][u] + hh[1][u]);
                    const float ng = fz_tanh(ih[2][u] + rg * hh[2][u]);
                    c[u] = 0.f;
                    h[u] = (1.f - zg) * ng + zg * sv[u];
                }
            }
            hv = make_float2(h[0], h[1]);
            if (blockIdx.x == 0) {
                if (lstm) *reinterpret_cast<float2*>(a.c_out + (size_t)b * H + (j2 << 1)) = make_float2(c[0], c[1]);
                if (a.h_out) *reinterpret_cast<float2*>(a.h_out + (size_t)b * H + (j2 << 1)) = hv;
            }
            if (a.x1) xv1 = r.x1;
        }
        reinterpret_cast<float2*>(xs)[e2] = hv;
        reinterpret_cast<float2*>(xs)[NB * H2 + e2] = xv1;
    };
    if (a.zero_h) {
        for (int e4 = tid; e4 < 2 * NB * H4; e4 += 256) reinterpret_cast<float4*>(xs)[e4] = f4_zero();
    } else {
        cell_finish(tid, cr);
        for (int e2 = tid + 256; e2 < NB * H2; e2 += 256) {
            cell_issue(e2, cr);
            cell_finish(e2, cr);
        }
    }
    __syncthreads();

    // ---- dot products of the four rows with the selected vector of every caption
    const float4* xv = reinterpret_cast<const float4*>(xs + (sg.xsel ? NB * H : 0));
    float acc[4][NB];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
    for (int base = 0; base < H4; base += 64 * IT) {
        if (base > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4* row = W4 + (size_t)min(row0 + r, sg.N - 1) * H4;
#pragma unroll
                for (int i = 0; i < IT; ++i) wv[r][i] = row[min(base + lane + 64 * i, H4 - 1)];
            }
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int k = base + lane + 64 * i;
            const int kc = min(k, H4 - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 w = k < H4 ? wv[r][i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[r][b] = dot4(w, xv[b * H4 + kc], acc[r][b]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float v = wave_sum(acc[r][b]);
            if (lane == 0) red[wave * 4 + r][b] = v;
        }
    __syncthreads();
    float v = -INFINITY;
    if (tid < FZ_ROWS * NB) {
        if (e_ok) {
            v = fmaf(red[er][eb], e_scale, e_bias);
            if (sg.y) sg.y[(size_t)eb * sg.ldy + en] = v;
        }
        red[er][eb] = v;
    }
    if (sg.amax) {
        __syncthreads();
        if (tid < nb) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < FZ_ROWS; ++r) {
                const int n = lblk * FZ_ROWS + r;
                if (n < sg.N) argmax_combine(bv, bi, red[r][tid], n);
            }
            const int nblk = (sg.N + FZ_ROWS - 1) / FZ_ROWS;
            a.pv_out[(size_t)tid * nblk + lblk] = bv;
            a.pi_out[(size_t)tid * nblk + lblk] = bi;
        }
    }
}

static int fz_blocks(int N) { return (N + FZ_ROWS - 1) / FZ_ROWS; }

static int fz_launch(FzArgs& a, int NB, hipStream_t st) {
    int blocks = 0;
    for (int s = 0; s < a.nseg; ++s) {
        a.blk0[s] = blocks;
        blocks += fz_blocks(a.seg[s].N);
    }
    a.blk0[a.nseg] = blocks;
    const int nbt = NB <= 1 ? 1 : (NB <= 2 ? 2 : (NB <= 4 ? 4 : 8));
    const size_t lds = (size_t)2 * nbt * a.H * sizeof(float);
    switch (nbt) {
        case 1: hipLaunchKernelGGL(fused_rows_kernel<1>, dim3(blocks), dim3(256), lds, st, a, NB); break;
        case 2: hipLaunchKernelGGL(fused_rows_kernel<2>, dim3(blocks), dim3(256), lds, st, a, NB); break;
        case 4: hipLaunchKernelGGL(fused_rows_kernel<4>, dim3(blocks), dim3(256), lds, st, a, NB); break;
        default: hipLaunchKernelGGL(fused_rows_kernel<8>, dim3(blocks), dim3(256), lds, st, a, NB); break;
    }
    return launch_status("greedy (fused steps)");
}

static inline long long fz_align4(long long n) { return (n + 3) & ~3LL; }

static int fz_check(const char* what, int NB, int H, int V, int steps) {
    if (NB <= 0 || H <= 0 || V <= 0 || steps <= 0) { set_error("%s: bad arguments", what); return ICAP_ERR_ARG; }
    if (NB > FZ_MAX_NB) { set_error("%s: NB=%d exceeds %d captions per call", what, NB, FZ_MAX_NB); return ICAP_ERR_LIMIT; }
    if (H % 4) { set_error("%s: H must be a multiple of 4", what); return ICAP_ERR_ALIGN; }
    if ((long long)NB * fz_blocks(V) > FZ_PART) { set_error("%s: NB x vocabulary blocks exceeds the staging buffer", what); return ICAP_ERR_LIMIT; }
    // the two input vectors of every caption slot live in LDS beside 33 KB of static staging: 32 KB of them at most
    const int nbt = NB <= 1 ? 1 : (NB <= 2 ? 2 : (NB <= 4 ? 4 : 8));
    if (2LL * nbt * H * 4 > 32 * 1024) { set_error("%s: NB=%d captions of width H=%d exceed the kernel's LDS", what, NB, H); return ICAP_ERR_LIMIT; }
    return ICAP_OK;
}

// ------------------------------------------------------------------ Bahdanau LSTMCell decoder: one launch per step
// workspace (floats): hh [2][NB][4H] | c [2][NB][H] | part_val [2][NB][P] | part_idx [2][NB][P]
long long greedy_bah_fused_ws_floats(int NB, int H, int V) {
    const long long P = fz_blocks(V);
    return 2 * fz_align4((long long)NB * 4 * H) + 2 * fz_align4((long long)NB * H) + 4 * fz_align4((long long)NB * P);
}

int greedy_bah_fused_f32(const float* g_tab, const float* g_ctx, const float* w_hh, const float* fc_w, const float* fc_scale,
                         const float* fc_bias, const long long* start_ids, long long* ids, float* logits_out, float* ws, int NB,
                         int H, int V, int steps, hipStream_t st) {
    if (!g_tab || !g_ctx || !w_hh || !fc_w || !start_ids || !ids || !ws) { set_error("greedy_bah_fused: bad arguments"); return ICAP_ERR_ARG; }
    FZ_TRY(fz_check("greedy_bah_fused", NB, H, V, steps));
    if (!aligned16(w_hh) || !aligned16(fc_w) || !aligned16(ws)) { set_error("greedy_bah_fused: 16-byte alignment required"); return ICAP_ERR_ALIGN; }
    const int P = fz_blocks(V);
    const long long hsz = fz_align4((long long)NB * 4 * H), csz = fz_align4((long long)NB * H), psz = fz_align4((long long)NB * P);
    float* hh = ws;
    float* cb = hh + 2 * hsz;
    float* pv = cb + 2 * csz;
    int* pi = reinterpret_cast<int*>(pv + 2 * psz);
    // h = c = 0 (init_hidden_state): the recurrent half of the first step's gates is zero
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(float) * (2 * hsz + 2 * csz), st);
    if (e != hipSuccess) { set_error("greedy_bah_fused: memset: %s", hipGetErrorString(e)); return ICAP_ERR_HIP_BASE + (int)e; }
    int cur = 0;
    for (int t = 0; t < steps; ++t) {
        FzArgs a;
        a.H = H; a.G = 4; a.V = V;
        a.tab = g_tab; a.add_a = g_ctx; a.add_b = hh + cur * hsz;
        a.c_in = cb + cur * csz; a.c_out = cb + (cur ^ 1) * csz;
        if (t == 0) a.tok_in = start_ids;
        else { a.pv_in = pv + cur * psz; a.pi_in = pi + cur * psz; a.nparts_in = P; a.ids_out = ids + (t - 1); a.ids_stride = steps; }
        a.pv_out = pv + (cur ^ 1) * psz; a.pi_out = pi + (cur ^ 1) * psz;
        a.nseg = 2;
        a.seg[0].W = fc_w; a.seg[0].scale = fc_scale; a.seg[0].bias = fc_bias; a.seg[0].N = V; a.seg[0].amax = 1;
        a.seg[0].y = logits_out ? logits_out + (size_t)t * NB * V : nullptr; a.seg[0].ldy = V;
        a.seg[1].W = w_hh; a.seg[1].N = 4 * H; a.seg[1].y = hh + (cur ^ 1) * hsz; a.seg[1].ldy = 4 * H;
        FZ_TRY(fz_launch(a, NB, st));
        cur ^= 1;
    }
    return argmax_finalize_f32(pv + cur * psz, pi + cur * psz, P, ids + (steps - 1), steps, nullptr, nullptr, nullptr, 0, 0, NB, st);
}

// ------------------------------------------------------------------ two-layer LSTM / GRU (seq2seq): two launches per step
// workspace (floats): hh0 [2][NB][GH] | hh1 [2][NB][GH] | ih1 [NB][GH] | h0 [2][NB][H] | c0 [2] | h1 [2] | c1 [2] | pv [2][NB][P] | pi [2]
long long greedy_rnn2_fused_ws_floats(int NB, int G, int H, int V) {
    const long long P = fz_blocks(V);
    return 5 * fz_align4((long long)NB * G * H) + 8 * fz_align4((long long)NB * H) + 4 * fz_align4((long long)NB * P);
}

// g_tab0 [V][G H]: layer 0's input side per token (W_ih0 . in(token) + its biases); g_first [NB][G H]: the same for the image
// feature that is step 0's input.  The recurrent rows carry hh*_bias (GRU: b_hh; LSTM: null, both biases sit on the input
// side), layer 1's input rows ih1_bias.
int greedy_rnn2_fused_f32(int is_lstm, const float* g_tab0, const float* g_first, const float* w_hh0, const float* hh0_bias,
                          const float* w_ih1, const float* ih1_bias, const float* w_hh1, const float* hh1_bias, const float* fc_w,
                          const float* fc_b, long long* ids, float* logits_out, float* ws, int NB, int H, int V, int steps,
                          hipStream_t st) {
    if (!g_tab0 || !g_first || !w_hh0 || !w_ih1 || !w_hh1 || !fc_w || !ids || !ws) { set_error("greedy_rnn2_fused: bad arguments"); return ICAP_ERR_ARG; }
    FZ_TRY(fz_check("greedy_rnn2_fused", NB, H, V, steps));
    if (!aligned16(w_hh0) || !aligned16(w_ih1) || !aligned16(w_hh1) || !aligned16(fc_w) || !aligned16(ws)) {
        set_error("greedy_rnn2_fused: 16-byte alignment required");
        return ICAP_ERR_ALIGN;
    }
    const int G = is_lstm ? 4 : 3, GH = G * H, P = fz_blocks(V);
    const long long gsz = fz_align4((long long)NB * GH), hsz = fz_align4((long long)NB * H), psz = fz_align4((long long)NB * P);
    float* hh0 = ws;
    float* hh1 = hh0 + 2 * gsz;
    float* ih1 = hh1 + 2 * gsz;
    float* h0 = ih1 + gsz;
    float* c0 = h0 + 2 * hsz;
    float* h1 = c0 + 2 * hsz;
    float* c1 = h1 + 2 * hsz;
    float* pv = c1 + 2 * hsz;
    int* pi = reinterpret_cast<int*>(pv + 2 * psz);
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(float) * (5 * gsz + 8 * hsz), st);          // hidden = None: zero state
    if (e != hipSuccess) { set_error("greedy_rnn2_fused: memset: %s", hipGetErrorString(e)); return ICAP_ERR_HIP_BASE + (int)e; }
    int cur = 0;
    if (hh0_bias || hh1_bias) {
        // recurrent side of a zero state = its bias: one launch of the recurrent rows on x = 0
        FzArgs a;
        a.H = H; a.G = G; a.zero_h = 1;
        a.nseg = 2;
        a.seg[0].W = w_hh1; a.seg[0].bias = hh1_bias; a.seg[0].N = GH; a.seg[0].y = hh1 + cur * gsz; a.seg[0].ldy = GH;
        a.seg[1].W = w_hh0; a.seg[1].bias = hh0_bias; a.seg[1].N = GH; a.seg[1].y = hh0 + cur * gsz; a.seg[1].ldy = GH; a.seg[1].xsel = 1;
        FZ_TRY(fz_launch(a, NB, st));
    }
    for (int t = 0; t < steps; ++t) {
        {   // layer 0's cell in the prologue, layer 1's input rows
            FzArgs a;
            a.H = H; a.G = G; a.V = V;
            if (t == 0) a.add_a = g_first;
            else { a.tab = g_tab0; a.pv_in = pv + cur * psz; a.pi_in = pi + cur * psz; a.nparts_in = P; a.ids_out = ids + (t - 1); a.ids_stride = steps; }
            a.add_b = hh0 + cur * gsz;
            a.c_in = c0 + cur * hsz; a.c_out = c0 + (cur ^ 1) * hsz;
            a.h_prev = h0 + cur * hsz; a.h_out = h0 + (cur ^ 1) * hsz;
            a.nseg = 1;
            a.seg[0].W = w_ih1; a.seg[0].bias = ih1_bias; a.seg[0].N = GH; a.seg[0].y = ih1; a.seg[0].ldy = GH;
            FZ_TRY(fz_launch(a, NB, st));
        }
        {   // layer 1's cell in the prologue; vocabulary projection and both layers' recurrent rows for the next step
            FzArgs a;
            a.H = H; a.G = G; a.V = V;
            a.add_a = ih1; a.add_b = hh1 + cur * gsz;
            a.c_in = c1 + cur * hsz; a.c_out = c1 + (cur ^ 1) * hsz;
            a.h_prev = h1 + cur * hsz; a.h_out = h1 + (cur ^ 1) * hsz;
            a.x1 = h0 + (cur ^ 1) * hsz;
            a.pv_out = pv + (cur ^ 1) * psz; a.pi_out = pi + (cur ^ 1) * psz;
            a.nseg = 3;
            a.seg[0].W = fc_w; a.seg[0].bias = fc_b; a.seg[0].N = V; a.seg[0].amax = 1;
            a.seg[0].y = logits_out ? logits_out + (size_t)t * NB * V : nullptr; a.seg[0].ldy = V;
            a.seg[1].W = w_hh1; a.seg[1].bias = hh1_bias; a.seg[1].N = GH; a.seg[1].y = hh1 + (cur ^ 1) * gsz; a.seg[1].ldy = GH;
            a.seg[2].W = w_hh0; a.seg[2].bias = hh0_bias; a.seg[2].N = GH; a.seg[2].y = hh0 + (cur ^ 1) * gsz; a.seg[2].ldy = GH; a.seg[2].xsel = 1;
            FZ_TRY(fz_launch(a, NB, st));
        }
        cur ^= 1;
    }
    return argmax_finalize_f32(pv + cur * psz, pi + cur * psz, P, ids + (steps - 1), steps, nullptr, nullptr, nullptr, 0, 0, NB, st);
}

}  // namespace icap
