// MBConv front half for the 14 x 14 / 7 x 7 stride-1 blocks of EfficientNet-B0, whole images per workgroup, with the expand
// GEMM as fp32-ACCURATE split products on the bf16 matrix cores and the depthwise convolution of one channel chunk issued
// BETWEEN the matrix instructions of the next chunk:
//
//   y = swish(bn(depthwise(swish(bn(x . We^T)))))   (+ the per-image channel sums squeeze-excite needs)
//
// On gfx950 matrix-core time and vector time of DIFFERENT waves of a SIMD add up, while vector instructions placed between
// the MFMAs of the SAME wave run in the matrix instruction's shadow.  So a workgroup (4 waves, one per SIMD, up to 512
// registers each) owns 196 pixels (one 14 x 14 image or four 7 x 7 images) and walks a range of channel chunks:
//
//   * the A operand never moves: each wave keeps the three bf16 planes (exact split hi + mid + lo of the fp32 value) of its
//     two 32-pixel row blocks in registers for all of K, split once when the workgroup starts;
//   * the B operand (expand weights) comes pre-split and pre-packed in fragment order (pack kernel below, once per weight),
//     so a K slice of 16 is six 16-byte loads per lane straight into registers, fetched one slice ahead - no LDS, no
//     barrier, no hand-counted vmcnt in the GEMM;
//   * per chunk: the six products per K slice (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi) accumulate chunk c + 1 while,
//     in the same basic block, the wave runs one piece of chunk c's depthwise work (a filter row of its output row, or the
//     BatchNorm + swish + store epilogue) from the LDS copy E of chunk c's expanded activations;
//   * then one barrier, E is rewritten with swish(bn(acc)) of chunk c + 1 (the image's rows sit between zero rows, so the
//     depthwise phase has no row bounds to test), one barrier, next chunk.
#pragma once
#include "common.hpp"
#include <utility>

namespace icap {

// 16-byte streaming store for the 14 x 14 layers: their expanded tensor (0.4-0.5 GB at B = 1024) is larger than the caches and is
// read next by another kernel; the 7 x 7 layers' (0.23 GB) is not, and plain stores are faster there
typedef float x3_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void x3_stream_store(float* dst, const float4& v) {
    __builtin_nontemporal_store(x3_f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<x3_f32x4*>(dst));
}


typedef float x3_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));

constexpr int X3_ROWS = 196;     // pixels per workgroup

#ifdef X3_STAMP      // profiling probe: cycles wave 0 of each workgroup spends per phase, summed over workgroups
__device__ unsigned long long x3_stamps[8];
#define X3_T(i) do { if (tid == 0) { const long long now_ = __builtin_amdgcn_s_memtime(); st_acc[i] += now_ - st_last; st_last = now_; } } while (0)
#else
#define X3_T(i) do { } while (0)
#endif

__device__ __forceinline__ unsigned x3_pk2(float a, float b) {       // (bf16(a), bf16(b)), a in the low half; round to nearest even
    const x3_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, x3_bf16x2));
}

// f[8] -> its exact three-way bf16 split, one packed plane each; `keep` = false gives zeros
__device__ __forceinline__ void x3_split8(const float (&f)[8], bool keep, uint4& hi, uint4& mid, uint4& lo) {
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = keep ? f[k] : 0.f;
    unsigned w[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            w[p][j] = x3_pk2(r[2 * j], r[2 * j + 1]);
            r[2 * j] -= __uint_as_float(w[p][j] << 16);                 // exact in fp32
            r[2 * j + 1] -= __uint_as_float(w[p][j] & 0xffff0000u);
        }
    }
    hi = make_uint4(w[0][0], w[0][1], w[0][2], w[0][3]);
    mid = make_uint4(w[1][0], w[1][1], w[1][2], w[1][3]);
    lo = make_uint4(w[2][0], w[2][1], w[2][2], w[2][3]);
}

// Only LDS traffic has to be complete at these barriers: a __syncthreads() would also wait for the output stores in flight.
__device__ __forceinline__ void x3_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// Expand weights [Cexp][CIN] fp32 -> packed planes [chunk][slice][plane][k half][XCH rows][8 bf16]: the 16 bytes lane
// (row, half) of an MFMA B fragment needs are contiguous, a (chunk, slice, plane, half) block is one 16-byte-per-lane load.
__global__ void x3_pack_we_kernel(const float* __restrict__ we, uint4* __restrict__ out, int Cexp, int cin, int xch, int nchunk) {
    const int nk = cin / 16;
    const long long total = (long long)nchunk * nk * 2 * xch;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(idx % xch);
        const int half = (int)((idx / xch) & 1);
        const int cs = (int)(idx / (2 * xch));            // chunk * nk + slice
        const int chunk = cs / nk, t = cs - chunk * nk;
        const int c = chunk * xch + row;
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = c < Cexp ? we[(size_t)c * cin + 16 * t + 8 * half + k] : 0.f;
        uint4 p[3];
        x3_split8(f, true, p[0], p[1], p[2]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) out[(((size_t)cs * 3 + pl) * 2 + half) * xch + row] = p[pl];
    }
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression in the body
template <typename F, int... I>
__device__ __forceinline__ void x3_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void x3_static_for(F&& f) { x3_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int CIN, int KS, int HW, int NCB>
__global__ __launch_bounds__(256, 1) void expand_dw_img_x3_kernel(const float* __restrict__ x, const uint4* __restrict__ we3p,
                                                                  const float* __restrict__ es, const float* __restrict__ eb,
                                                                  const float* __restrict__ wd, const float* __restrict__ ds,
                                                                  const float* __restrict__ db, float* __restrict__ y,
                                                                  float* __restrict__ pooled, int B, int Cexp, int nchunk, int cpg,
                                                                  int nsplit) {
    static_assert(HW == 14 || HW == 7, "whole-image tiles: 14 x 14 (one image) or 7 x 7 (four images)");
    static_assert(CIN % 16 == 0 && (KS == 3 || KS == 5) && (NCB == 1 || NCB == 2), "K slices of 16; 3x3 or 5x5; 32 or 64 channels per chunk");
    constexpr int IMGS = HW == 14 ? 1 : 4, PX = HW * HW, NK = CIN / 16, PAD = KS / 2, RPI = HW + 2 * PAD;
    constexpr int XCH = 32 * NCB, NQ = XCH / 4, RSLOTS = 256 / NQ, NT = IMGS * HW;
    // E pixel pitch: for 64 channels exactly the 256 bytes of the LDS bank row, so the four-channel slices of ANY two pixels
    // meet in a bank only for equal channels, and a 16-byte read's lane groups (which mix two depthwise tasks) never do;
    // for 32 channels 36 floats
    constexpr int ES = NCB == 2 ? XCH : XCH + 4;
    constexpr int PD = 3;                                   // LDS reads of E a thread keeps in flight ahead of their use
    static_assert(NT <= RSLOTS, "one output row per thread");
    constexpr int WROWS = KS * KS + 2;                      // filter taps, folded-BN scale, folded-BN bias
    constexpr int NPW = (WROWS * NQ + 255) / 256;           // float4 parameter entries per thread
    constexpr int E_FLOATS = IMGS * RPI * HW * ES;
    constexpr int NITEMS = KS + 1;                          // depthwise pieces of a chunk: KS filter rows, then the epilogue
    __shared__ __attribute__((aligned(16))) float E[E_FLOATS];
    __shared__ float4 wds[2][WROWS][NQ];
    __shared__ __attribute__((aligned(16))) float red[NT][XCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    // XCD x walks a contiguous range of the (image group, chunk range) list, range fastest: the ranges of one group share an L2
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    const int t_lin = xcd * (nwg >> 3) + (xcd < (nwg & 7) ? xcd : (nwg & 7)) + slot0;
    const int grp = t_lin / nsplit, split = t_lin - grp * nsplit;
    const int img0 = grp * IMGS;
    const int c0 = split * cpg, c1 = min(nchunk, c0 + cpg);
    if (c0 >= c1) return;                                   // workgroup-uniform

    // ---- depthwise task of this thread: one output row (all its columns) of one image, four channels
    const int q = tid % NQ, rslot = tid / NQ;
    const bool tvalid = rslot < NT;
    const int task = tvalid ? rslot : 0;
    const int timg = task / HW, trow = task - timg * HW;
    const bool img_ok = img0 + timg < B;

    // ---- E: zero everywhere once (the rows above and below every image stay zero for the workgroup's life)
    for (int idx = tid; idx < E_FLOATS / 4; idx += 256) reinterpret_cast<float4*>(E)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- per-chunk parameters: depthwise taps / scale / bias as float4 entries, expand scale / bias per accumulator column
    float4 pw[NPW];
    auto load_dw_params = [&](int chunk) {
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int idx = tid + 256 * n;
            const int row = idx / NQ, qq = idx - row * NQ;
            const int c = chunk * XCH + qq * 4;
            const bool ok = row < WROWS && c < Cexp;
            const float* src = row < KS * KS ? wd + (size_t)row * Cexp : (row == KS * KS ? ds : db);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row == KS * KS && !ds) v = make_float4(1.f, 1.f, 1.f, 1.f);
            if (ok && src) v = *reinterpret_cast<const float4*>(src + c);
            pw[n] = v;
        }
    };
    auto store_dw_params = [&](int buf) {
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int idx = tid + 256 * n;
            if (idx < WROWS * NQ) (&wds[buf][0][0])[idx] = pw[n];
        }
    };
    float esc[NCB], ebc[NCB];
    auto load_expand_params = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            const int c = chunk * XCH + 32 * j + lr;
            const bool ok = c < Cexp;
            esc[j] = (ok && es) ? es[c] : 1.f;
            ebc[j] = (ok && eb) ? eb[c] : 0.f;
        }
    };

    // ---- A fragments: rows (2 wave + i) * 32 + lr, K slice t, this lane's half of the slice; split once, kept for all chunks
    uint4 a16[2][NK][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (2 * wave + i) * 32 + lr;
        const bool ok = row < X3_ROWS && img0 + row / PX < B;
        const float* src = x + ((size_t)img0 * PX + (ok ? row : 0)) * CIN + 8 * lh;
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            const float4 v0 = *reinterpret_cast<const float4*>(src + 16 * t), v1 = *reinterpret_cast<const float4*>(src + 16 * t + 4);
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            x3_split8(f, ok, a16[i][t][0], a16[i][t][1], a16[i][t][2]);
            // only matrix instructions read these: keep them in the accumulator half of the register file, so that the
            // vector half is left to the depthwise stage
#pragma unroll
            for (int p = 0; p < 3; ++p)
                asm volatile("" : "+a"(a16[i][t][p].x), "+a"(a16[i][t][p].y), "+a"(a16[i][t][p].z), "+a"(a16[i][t][p].w));
        }
    }

    // ---- B fragments of (chunk, slice): [column block][plane], one 16-byte load each
    auto load_b = [&](int chunk, int t, uint4 (&bf)[NCB][3]) {
        const uint4* src = we3p + ((size_t)(chunk * NK + t) * 6 + lh) * XCH + lr;
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[j][p] = src[p * 2 * XCH + 32 * j];
    };

    x3_f32x16 acc[2][NCB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NCB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    // matrix instruction m of K slice t: product pr (smallest terms first), row block i, column block j.  The empty asm
    // statements of this kernel are anchors: instruction selection orders only what a chain ties down, a volatile asm is
    // on the chain, and a value that passes through one cannot be computed later (or its consumers earlier) than the
    // asm stands - that is what holds the MFMA / depthwise interleave in the order written here.
    auto mfma_one = [&](int t, int m, const uint4 (&bf)[NCB][3]) {
        constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
        const int j = m % NCB, i = (m / NCB) % 2, pr = m / (2 * NCB);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3_bf16x8, a16[i][t][pa[pr]]),
                                                            __builtin_bit_cast(x3_bf16x8, bf[j][pb[pr]]), acc[i][j], 0, 0, 0);
        asm volatile("" : "+a"(acc[i][j]));
    };
    constexpr int NM = 12 * NCB;                            // matrix instructions per K slice

    // ---- depthwise work of the chunk whose activations are in E (parameters in wds[buf]), cut into steps of about a dozen
    // vector instructions; o, wr, dsv, dbv, ps carry over the steps.  Item k < KS is filter row k (step 0 fetches its taps,
    // step 1 + cc takes input column cc: one LDS read feeding the outputs it touches); item KS is the epilogue (step 1 + c
    // finishes output column c; the last step stores the row and its channel sums).
    constexpr int STEPS = HW + 1;
    float4 o[HW], wr[KS], evq[PD], dsv, dbv, ps;
    auto dw_step = [&](int item, int step, int buf, int chunk) {
        if (item < KS) {
            const int dr = item;
            if (step == 0) {
                if (dr == 0) {
#pragma unroll
                    for (int c = 0; c < HW; ++c) o[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int dc = 0; dc < KS; ++dc) wr[dc] = wds[buf][dr * KS + dc][q];
#pragma unroll
                for (int d = 0; d < PD; ++d) evq[d] = *reinterpret_cast<const float4*>(&E[((timg * RPI + trow + dr) * HW + d) * ES + q * 4]);
                return;
            }
            // the value of this step was fetched PD steps ago, this step fetches the one PD steps ahead: an LDS read under
            // load takes longer than one step's twenty FMAs
            const int cc = step - 1;
            const float4 ev = evq[cc % PD];
            if (cc + PD < HW) evq[cc % PD] = *reinterpret_cast<const float4*>(&E[((timg * RPI + trow + dr) * HW + cc + PD) * ES + q * 4]);
#pragma unroll
            for (int dc = 0; dc < KS; ++dc) {
                const int c = cc - dc + PAD;
                if (c >= 0 && c < HW) {
                    o[c].x = fmaf(ev.x, wr[dc].x, o[c].x);
                    o[c].y = fmaf(ev.y, wr[dc].y, o[c].y);
                    o[c].z = fmaf(ev.z, wr[dc].z, o[c].z);
                    o[c].w = fmaf(ev.w, wr[dc].w, o[c].w);
                    asm volatile("" : "+v"(o[c].x), "+v"(o[c].y), "+v"(o[c].z), "+v"(o[c].w));
                }
            }
        } else {
            if (step == 0) {
                dsv = wds[buf][KS * KS][q];
                dbv = wds[buf][KS * KS + 1][q];
                ps = make_float4(0.f, 0.f, 0.f, 0.f);
                return;
            }
            const int c = step - 1;
            float4 v;
            v.x = fmaf(o[c].x, dsv.x, dbv.x); v.y = fmaf(o[c].y, dsv.y, dbv.y); v.z = fmaf(o[c].z, dsv.z, dbv.z); v.w = fmaf(o[c].w, dsv.w, dbv.w);
            v.x *= sigmoid_fast(v.x); v.y *= sigmoid_fast(v.y); v.z *= sigmoid_fast(v.z); v.w *= sigmoid_fast(v.w);
            o[c] = v;
            ps.x += v.x; ps.y += v.y; ps.z += v.z; ps.w += v.w;
            asm volatile("" : "+v"(o[c].x), "+v"(o[c].y), "+v"(o[c].z), "+v"(o[c].w), "+v"(ps.x), "+v"(ps.y), "+v"(ps.z), "+v"(ps.w));
            if (step == HW) {
                const int cb = chunk * XCH + q * 4;
                const bool st_ok = tvalid && img_ok && cb < Cexp;
                float* dst = y + (((size_t)(img0 + timg) * PX + trow * HW) * Cexp + (cb < Cexp ? cb : 0));
                if (st_ok) {
#pragma unroll
                    for (int c2 = 0; c2 < HW; ++c2) *reinterpret_cast<float4*>(dst + (size_t)c2 * Cexp) = o[c2];   // 7 x 7 layers: the tensor fits the last-level cache, plain stores
                }
                if (tvalid) *reinterpret_cast<float4*>(&red[task][q * 4]) = ps;
            }
        }
    };

    // per-image channel sums of the chunk whose row sums are in red: the image's rows in order
    auto pooled_reduce = [&](int chunk) {
        if (!pooled) return;
        for (int idx = tid; idx < IMGS * XCH; idx += 256) {
            const int img = idx / XCH, c = idx - img * XCH;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < HW; ++r) s += red[img * HW + r][c];
            const int cg = chunk * XCH + c;
            if (cg < Cexp && img0 + img < B) pooled[(size_t)(img0 + img) * Cexp + cg] = s;
        }
    };
    // E = swish(bn(acc)) for the real pixels (row block 6 holds four, row block 7 none), accumulators cleared
    auto e_write = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rb = 2 * wave + i;                     // wave-uniform
            if (rb * 32 < X3_ROWS) {
                const bool full = rb * 32 + 32 <= X3_ROWS;
#pragma unroll
                for (int j = 0; j < NCB; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        const int pos = HW == 14 ? row + PAD * HW : row + PAD * HW * (2 * (row / PX) + 1);
                        const float v = fmaf(acc[i][j][e], esc[j], ebc[j]);
                        // the last row block holds four pixels: rows 192..195 are e = 0..3 of the lower lane half
                        if (full) E[pos * ES + 32 * j + lr] = v * sigmoid_fast(v);
                        else if (e < 4 && lh == 0) E[pos * ES + 32 * j + lr] = v * sigmoid_fast(v);
                    }
            }
        }
        zero_acc();
    };

    // ---- first chunk: GEMM alone
    uint4 bcur[NCB][3], bnxt[NCB][3];
    load_dw_params(c0);
    load_expand_params(c0);
    load_b(c0, 0, bcur);
    zero_acc();
#pragma unroll
    for (int t = 0; t < NK; ++t) {
        if (t + 1 < NK) load_b(c0, t + 1, bnxt);
        else load_b(min(c0 + 1, c1 - 1), 0, bnxt);
#pragma unroll
        for (int m = 0; m < NM; ++m) mfma_one(t, m, bcur);
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bcur[j][p] = bnxt[j][p];
    }
    x3_lds_barrier();                   // E is zero everywhere
    store_dw_params(0);
    e_write();
    x3_lds_barrier();

    // ---- steady state: GEMM of chunk c + 1 with the depthwise pieces of chunk c between its matrix instructions
    for (int c = c0; c + 1 < c1; ++c) {
        const int buf = (c - c0) & 1;
        load_dw_params(c + 1);
        load_expand_params(c + 1);
        x3_static_for<NK>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if (t + 1 < NK) load_b(c + 1, t + 1, bnxt);
            else load_b(min(c + 2, c1 - 1), 0, bnxt);
            // the depthwise items that run beside this slice: items [k0, k1) of the chunk, their steps spread evenly behind
            // the slice's matrix instructions
            constexpr int k0 = (t * NITEMS + NK - 1) / NK, k1 = ((t + 1) * NITEMS + NK - 1) / NK;
            constexpr int S = (k1 - k0) * STEPS;
            x3_static_for<NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                mfma_one(t, m, bcur);
                if constexpr (S > 0) {
                    x3_static_for<S>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        if constexpr (g * NM / S == m) dw_step(k0 + g / STEPS, g % STEPS, buf, c);
                    });
                }
            });
            __builtin_amdgcn_sched_barrier(0);                          // nothing moves across slices
#pragma unroll
            for (int j = 0; j < NCB; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) bcur[j][p] = bnxt[j][p];
        });
        x3_lds_barrier();               // every thread has read E and written its row sums
        pooled_reduce(c);
        store_dw_params(buf ^ 1);
        e_write();
        x3_lds_barrier();
    }

    // ---- last chunk: depthwise alone
    {
        const int c = c1 - 1, buf = (c - c0) & 1;
#pragma unroll
        for (int k = 0; k < NITEMS; ++k)
#pragma unroll
            for (int st = 0; st < STEPS; ++st) dw_step(k, st, buf, c);
        x3_lds_barrier();
        pooled_reduce(c);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The 14 x 14 layers with EIGHT waves (two per SIMD).  One wave per SIMD issues a vector instruction every four cycles at
// best and has nobody to cover its stalls; two waves issue one every two cycles between them and cover each other.  With
// 256 registers per wave the stationary A planes of TWO row blocks no longer fit, so a wave owns ONE 32-pixel row block
// (wave 7's block is empty: it only joins the barriers) and keeps the planes of the first NKR K slices in registers, the
// rest in LDS (written once, read back by the same wave).  The depthwise tasks are HALF rows (7 outputs): a thread of the
// right half works on the mirrored row - column 13 - u, tap KS - 1 - j - so both halves run the same instruction stream.
//
// STRIDE = 2 (the 14 x 14 -> 7 x 7 block): same GEMM and E tile, TF-'same' padding (PT rows / columns of zeros before the
// image).  The 49 outputs of a channel quad are spread over the same 28 task slots: wave w owns output row w, its four
// slots two neighbouring outputs each (the last slot one), over the seven input columns those need; a column outside the
// image reads a zero row of E instead.
template <int CIN, int KS, int NKR, int STRIDE = 1>
__global__ __launch_bounds__(512, 1) void expand_dw_img14_w8_kernel(const float* __restrict__ x, const uint4* __restrict__ we3p,
                                                                    const float* __restrict__ es, const float* __restrict__ eb,
                                                                    const float* __restrict__ wd, const float* __restrict__ ds,
                                                                    const float* __restrict__ db, float* __restrict__ y,
                                                                    float* __restrict__ pooled, int B, int Cexp, int nchunk, int cpg,
                                                                    int nsplit) {
    constexpr int HW = 14, PX = HW * HW, NK = CIN / 16, PAD = KS / 2, RPI = HW + 2 * PAD, NCB = 2;
    constexpr int HO = STRIDE == 2 ? HW / 2 : HW;                                           // output rows = columns
    constexpr int PT = STRIDE == 2 ? ((HO - 1) * 2 + KS - HW) / 2 : PAD;                    // zero rows / columns before the image
    constexpr int XCH = 64, ES = XCH, NQ = 16, HALF = 7;                                    // ES: see the four-wave kernel
    constexpr int NT = 2 * HW;                                                              // depthwise task slots
    constexpr int NOUT = STRIDE == 2 ? 2 : HALF;                                            // outputs per task
    constexpr int NIN = STRIDE == 2 ? (NOUT - 1) * 2 + KS : HALF + PAD;                     // input columns per task
    static_assert(STRIDE == 1 || STRIDE == 2, "stride");
    constexpr int PD = (CIN == 112 && KS == 5 && STRIDE == 1) ? 1 : 2;   // LDS reads of E in flight ahead of their use (one where two would spill)
    constexpr int NKL = NK > NKR ? NK - NKR : 0;            // K slices whose A planes live in LDS
    constexpr int WROWS = KS * KS + 2;
    static_assert(WROWS * NQ <= 512, "one float4 parameter entry per thread");
    constexpr int E_FLOATS = RPI * HW * ES;
    constexpr int NITEMS = KS + 1;
    __shared__ __attribute__((aligned(16))) float E[E_FLOATS];
    __shared__ float4 wds[2][WROWS][NQ];
    __shared__ __attribute__((aligned(16))) float red[NT][XCH];
    __shared__ uint4 Al[NKL ? NKL : 1][3][2][256];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
    const int t_lin = xcd * (nwg >> 3) + (xcd < (nwg & 7) ? xcd : (nwg & 7)) + slot0;
    const int img0 = t_lin / nsplit, split = t_lin - img0 * nsplit;
    const int c0 = split * cpg, c1 = min(nchunk, c0 + cpg);
    if (c0 >= c1) return;
#ifdef X3_STAMP
    long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    const bool img_ok = img0 < B;
    const bool works = wave < 7;                            // wave-uniform: row block 7 holds no pixel, slots 28..31 no task

    // depthwise task: output row trow, half th (0: columns 0..6, 1: columns 13..7 in that order), four channels
    const int q = tid % NQ, slot = tid / NQ;
    const bool tvalid = slot < NT;
    const int task = tvalid ? slot : 0;
    // stride 1: a wave's four tasks are consecutive rows of one half; stride 2: output row task / 4, outputs 2 * cg, 2 * cg + 1
    const int th = STRIDE == 2 ? 0 : task / HW, trow = STRIDE == 2 ? task >> 2 : task - th * HW, cg = task & 3;

    for (int idx = tid; idx < E_FLOATS / 4; idx += 512) reinterpret_cast<float4*>(E)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 pw;
    auto load_dw_params = [&](int chunk) {
        const int row = tid / NQ, qq = tid - row * NQ;
        const int c = chunk * XCH + qq * 4;
        const bool ok = row < WROWS && c < Cexp;
        const float* src = row < KS * KS ? wd + (size_t)row * Cexp : (row == KS * KS ? ds : db);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row == KS * KS && !ds) v = make_float4(1.f, 1.f, 1.f, 1.f);
        if (ok && src) v = *reinterpret_cast<const float4*>(src + c);
        pw = v;
    };
    auto store_dw_params = [&](int buf) {
        if (tid < WROWS * NQ) (&wds[buf][0][0])[tid] = pw;
    };
    float esc[NCB], ebc[NCB];
    auto load_expand_params = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            const int c = chunk * XCH + 32 * j + lr;
            const bool ok = c < Cexp;
            esc[j] = (ok && es) ? es[c] : 1.f;
            ebc[j] = (ok && eb) ? eb[c] : 0.f;
        }
    };

    // ---- A: row wave * 32 + lr, this lane's half of every K slice, split once
    uint4 a16[NKR < NK ? NKR : NK][3];
    {
        const int row = wave * 32 + lr;
        const bool ok = row < X3_ROWS && img_ok;
        const float* src = x + ((size_t)(img_ok ? img0 : 0) * PX + (ok ? row : 0)) * CIN + 8 * lh;
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            const float4 v0 = *reinterpret_cast<const float4*>(src + 16 * t), v1 = *reinterpret_cast<const float4*>(src + 16 * t + 4);
            const float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            uint4 pl[3];
            x3_split8(f, ok, pl[0], pl[1], pl[2]);
            if (t < NKR) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a16[t][p] = pl[p];
                    // anchored in ordinary registers: with two waves per SIMD the compiler splits the 256 registers evenly
                    // between the two classes as soon as one value asks for the accumulator class, and 128 are too few here
                    asm volatile("" : "+v"(a16[t][p].x), "+v"(a16[t][p].y), "+v"(a16[t][p].z), "+v"(a16[t][p].w));
                }
            } else {
#pragma unroll
                for (int p = 0; p < 3; ++p) Al[t - NKR][p][lh][wave * 32 + lr] = pl[p];     // read back by this wave only
            }
        }
    }

    auto load_b = [&](int chunk, int t, uint4 (&bf)[NCB][3]) {
        const uint4* src = we3p + ((size_t)(chunk * NK + t) * 6 + lh) * XCH + lr;
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[j][p] = src[p * 2 * XCH + 32 * j];
    };
    // A planes of slice t for the matrix instructions: the registers, or this wave's rows of the LDS copy
    auto load_a = [&](int t, uint4 (&af)[3]) {
        if (t >= NKR) {
#pragma unroll
            for (int p = 0; p < 3; ++p) af[p] = Al[t - NKR][p][lh][wave * 32 + lr];
        }
    };

    x3_f32x16 acc[NCB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NCB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    };
    constexpr int NM = 6 * NCB;                             // matrix instructions per K slice and wave
    auto mfma_one = [&](int t, int m, const uint4 (&af)[3], const uint4 (&bf)[NCB][3]) {
        constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
        const int j = m % NCB, pr = m / NCB;
        const uint4 a = t < NKR ? a16[t < NKR ? t : 0][pa[pr]] : af[pa[pr]];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3_bf16x8, a), __builtin_bit_cast(x3_bf16x8, bf[j][pb[pr]]),
                                                         acc[j], 0, 0, 0);
        asm volatile("" : "+v"(acc[j]));
    };

    // ---- depthwise steps (see the four-wave kernel): item k < KS = filter row k (step 0: taps; step 1 + u: input u of the
    // mirrored half row), item KS = epilogue (step 1 + l: output l; the last step stores)
    constexpr int ROW_STEPS = NIN + 1, EPI_STEPS = NOUT + 1;
    float4 o[NOUT], wr[KS], evq[PD], dsv, dbv, ps;
    auto ecol = [&](int u) { return th ? HW - 1 - u : u; };
    // input u of the task in filter row erow: stride 2 - column 4 * cg - PT + u, or E's first (all-zero) row outside the image
    auto eload = [&](const float* erow, int u) {
        if constexpr (STRIDE == 2) {
            const int col = 4 * cg - PT + u;
            return *reinterpret_cast<const float4*>((unsigned)col < (unsigned)HW ? erow + col * ES : &E[q * 4]);
        } else {
            return *reinterpret_cast<const float4*>(erow + ecol(u) * ES);
        }
    };
    // guarded = false: the caller knows that every working lane stores (steady-state chunks of the stride-1 form are whole, the
    // image exists, wave 7 is not here) - the stores then sit in straight-line code, the compiler can count them, and a later
    // wait for a load issued before them need not wait for them
    auto dw_step = [&](int item, int step, int buf, int chunk, bool guarded = true) {
        if (item < KS) {
            const int dr = item;
            const float* erow = &E[((STRIDE * trow + dr + PAD - PT) * HW) * ES + q * 4];
            if (step == 0) {
                if (dr == 0) {
#pragma unroll
                    for (int l = 0; l < NOUT; ++l) o[l] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int j = 0; j < KS; ++j) wr[j] = wds[buf][dr * KS + (th ? KS - 1 - j : j)][q];
#pragma unroll
                for (int d = 0; d < PD; ++d) evq[d] = eload(erow, d);
                return;
            }
            const int u = step - 1;
            const float4 ev = evq[u % PD];
            if (u + PD < NIN) evq[u % PD] = eload(erow, u + PD);
#pragma unroll
            for (int l = 0; l < NOUT; ++l) {
                const int j = STRIDE == 2 ? u - 2 * l : u - l + PAD;   // (mirrored) tap feeding output l from input u
                if (j >= 0 && j < KS) {
                    o[l].x = fmaf(ev.x, wr[j].x, o[l].x);
                    o[l].y = fmaf(ev.y, wr[j].y, o[l].y);
                    o[l].z = fmaf(ev.z, wr[j].z, o[l].z);
                    o[l].w = fmaf(ev.w, wr[j].w, o[l].w);
                    asm volatile("" : "+v"(o[l].x), "+v"(o[l].y), "+v"(o[l].z), "+v"(o[l].w));
                }
            }
        } else {
            if (step == 0) {
                dsv = wds[buf][KS * KS][q];
                dbv = wds[buf][KS * KS + 1][q];
                ps = make_float4(0.f, 0.f, 0.f, 0.f);
                return;
            }
            const int l = step - 1;
            float4 v;
            v.x = fmaf(o[l].x, dsv.x, dbv.x); v.y = fmaf(o[l].y, dsv.y, dbv.y); v.z = fmaf(o[l].z, dsv.z, dbv.z); v.w = fmaf(o[l].w, dsv.w, dbv.w);
            v.x *= sigmoid_fast(v.x); v.y *= sigmoid_fast(v.y); v.z *= sigmoid_fast(v.z); v.w *= sigmoid_fast(v.w);
            if constexpr (STRIDE == 2) {   // the last slot of a row has one output: its second one counts as zero
                const float keep = 2 * cg + l < HO ? 1.f : 0.f;
                v.x *= keep; v.y *= keep; v.z *= keep; v.w *= keep;
            }
            o[l] = v;
            ps.x += v.x; ps.y += v.y; ps.z += v.z; ps.w += v.w;
            asm volatile("" : "+v"(o[l].x), "+v"(o[l].y), "+v"(o[l].z), "+v"(o[l].w), "+v"(ps.x), "+v"(ps.y), "+v"(ps.z), "+v"(ps.w));
            if (step == NOUT) {
                const int cb = chunk * XCH + q * 4;
                const bool st_ok = !guarded || (tvalid && img_ok && cb < Cexp);
                const size_t dst_off = ((size_t)(img_ok ? img0 : 0) * (HO * HO) + trow * HO) * Cexp + (cb < Cexp ? cb : 0);
                float* dst = y + dst_off;
                if (st_ok) {
                    if constexpr (STRIDE == 2) {   // 7 x 7 output: cache resident, plain stores
#pragma unroll
                        for (int l2 = 0; l2 < NOUT; ++l2)
                            if (2 * cg + l2 < HO) *reinterpret_cast<float4*>(dst + (size_t)(2 * cg + l2) * Cexp) = o[l2];
                    } else {
                        // a running pointer (one step per output, backwards for the mirrored half): seven precomputed 64-bit
                        // pixel offsets would be spilled, and a spill reload between two stores waits for the first store
                        size_t off = dst_off + (th ? (size_t)(HW - 1) * Cexp : (size_t)0);
                        const long long step = th ? -(long long)Cexp : (long long)Cexp;
#pragma unroll
                        for (int l2 = 0; l2 < HALF; ++l2) {
                            x3_stream_store(y + off, o[l2]);
                            off += step;
                            asm volatile("" : "+v"(off));      // keep it a running offset
                        }
                    }
                }
                if (tvalid) *reinterpret_cast<float4*>(&red[task][q * 4]) = ps;
            }
        }
    };

    auto pooled_reduce = [&](int chunk) {
        if (!pooled || tid >= XCH) return;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < HW; ++r) s += red[r][tid] + red[HW + r][tid];      // all task slots in a fixed order (stride 1: rows in order, left half + right half)
        const int cg = chunk * XCH + tid;
        if (cg < Cexp && img_ok) pooled[(size_t)img0 * Cexp + cg] = s;
    };
    auto e_write = [&]() {
        const int rb = wave;
        if (rb * 32 < X3_ROWS) {
            const bool full = rb * 32 + 32 <= X3_ROWS;
#pragma unroll
            for (int j = 0; j < NCB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    const float v = fmaf(acc[j][e], esc[j], ebc[j]);
                    if (full) E[(row + PAD * HW) * ES + 32 * j + lr] = v * sigmoid_fast(v);
                    else if (e < 4 && lh == 0) E[(row + PAD * HW) * ES + 32 * j + lr] = v * sigmoid_fast(v);
                }
        }
        zero_acc();
    };

    // ---- first chunk: GEMM alone
    uint4 bcur[NCB][3], bnxt[NCB][3], acur[3], anxt[3];
    X3_T(0);                 // prologue: E zero, A load + split
    load_dw_params(c0);
    load_expand_params(c0);
    zero_acc();
    if (works) {
        load_b(c0, 0, bcur);
        load_a(0, acur);
        x3_static_for<NK>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if (t + 1 < NK) load_b(c0, t + 1, bnxt);
            else load_b(min(c0 + 1, c1 - 1), 0, bnxt);
            load_a((t + 1) % NK, anxt);
#pragma unroll
            for (int m = 0; m < NM; ++m) mfma_one(t, m, acur, bcur);
#pragma unroll
            for (int j = 0; j < NCB; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) bcur[j][p] = bnxt[j][p];
#pragma unroll
            for (int p = 0; p < 3; ++p) acur[p] = anxt[p];
        });
    }
    X3_T(1);                 // first GEMM
    x3_lds_barrier();
    store_dw_params(0);
    e_write();
    x3_lds_barrier();
    X3_T(2);                 // first E write + barriers

    for (int c = c0; c + 1 < c1; ++c) {
        const int buf = (c - c0) & 1;
        load_dw_params(c + 1);
        load_expand_params(c + 1);
        if (works) {
            x3_static_for<NK>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if (t + 1 < NK) load_b(c + 1, t + 1, bnxt);
                else load_b(min(c + 2, c1 - 1), 0, bnxt);
                load_a((t + 1) % NK, anxt);
                constexpr int k0 = (t * NITEMS + NK - 1) / NK, k1 = ((t + 1) * NITEMS + NK - 1) / NK;
                if constexpr (t == NK - 1) {
                    // the next chunk's parameters (requested at the top of this chunk) take their place before the last slice,
                    // i.e. before this chunk's output stores are issued: behind the stores the wait for these loads would be a
                    // wait for the stores as well.  wds[buf ^ 1] is free: its readers were the previous chunk's steps.
                    store_dw_params(buf ^ 1);
#pragma unroll
                    for (int j = 0; j < NCB; ++j) asm volatile("" : "+v"(esc[j]), "+v"(ebc[j]));
                }
                // steps of the items [k0, k1) that run beside this slice, in order: (item, step) pairs numbered g = 0 .. S - 1
                constexpr int S = (k1 > k0 ? (k1 - k0 - (k1 == NITEMS ? 1 : 0)) * ROW_STEPS + (k1 == NITEMS ? EPI_STEPS : 0) : 0);
                x3_static_for<NM>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    mfma_one(t, m, acur, bcur);
                    if constexpr (S > 0) {
                        x3_static_for<S>([&](auto gc) {
                            constexpr int g = decltype(gc)::value;
                            if constexpr (g * NM / S == m) {
                                constexpr int item = k0 + g / ROW_STEPS;        // only the chunk's last item is shorter
                                dw_step(item, g - (item - k0) * ROW_STEPS, buf, c, STRIDE != 1);
                            }
                        });
                    }
                });
#pragma unroll
                for (int j = 0; j < NCB; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bcur[j][p] = bnxt[j][p];
#pragma unroll
                for (int p = 0; p < 3; ++p) acur[p] = anxt[p];
            });
        }
        X3_T(3);             // steady slices
        x3_lds_barrier();
        X3_T(4);             // barrier after the slices
        pooled_reduce(c);
        e_write();
        X3_T(5);             // pooled + E write
        x3_lds_barrier();
        X3_T(6);             // barrier after the E write
    }

    {
        const int c = c1 - 1, buf = (c - c0) & 1;
        if (works) {
#pragma unroll
            for (int k = 0; k < NITEMS; ++k)
#pragma unroll
                for (int st = 0; st < (k < KS ? ROW_STEPS : EPI_STEPS); ++st) dw_step(k, st, buf, c);
        }
        x3_lds_barrier();
        pooled_reduce(c);
    }
    X3_T(7);                 // drain
#ifdef X3_STAMP
    if (tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&x3_stamps[i], (unsigned long long)st_acc[i]);
#endif
}

}  // namespace icap
