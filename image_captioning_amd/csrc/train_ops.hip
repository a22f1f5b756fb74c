// Row / column / element kernels of the training-mode forward and of the backward pass:
// column reductions (bias gradients, BatchNorm statistics and gradients, LayerNorm
// parameter gradients), BatchNorm and LayerNorm in training mode, dropout, activation
// gradients, the LSTM cell gradient, the additive-attention gradient and the embedding
// gradient.  All of them are bound by memory traffic; reductions run in a fixed order
// (two passes, no atomics), so a training step is reproducible bit for bit.
#include <initializer_list>
#include "common.hpp"

namespace icap {

// ---------------------------------------------------------------- column reductions
// x, y: [M][C] row-major (row strides ldx / ldy).  A workgroup owns 64 columns and one chunk of
// COL_ROWS rows; partial sums go to part[chunk][C] (two planes when the mode has two results) and
// col_finish_kernel adds the chunks in order.
constexpr int COL_ROWS = 512;

enum ColMode : int {
    COL_SUM = 0,        // s1 = sum a
    COL_SQDEV = 1,      // s1 = sum (a - v1[c])^2                                (variance given the mean)
    COL_BN_BWD = 2,     // s1 = sum g, s2 = sum g * (x - v1[c]) * v2[c]          (g = a * act'(x*v3[c] + v4[c]))
    COL_LN_BWD = 3,     // s1 = sum a, s2 = sum a * (x - r1[r]) * r2[r]          (per-row mean / rstd)
    COL_MOMENTS = 4,    // s1 = sum (a - v1[c]), s2 = sum (a - v1[c])^2          (mean and variance in one pass, v1 = a shift near the mean)
};

__device__ __forceinline__ float act_grad(float z, int act) {
    switch (act) {
        case ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case ACT_SWISH: { const float s = sigmoid_fast(z); return s * (1.f + z * (1.f - s)); }
        case ACT_SIGMOID: { const float s = sigmoid_fast(z); return s * (1.f - s); }
        case ACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
        default: return 1.f;
    }
}

// A gradient that still has to pass a squeeze-excite gate on its way into a BatchNorm: the kernels that take this read
//   a_eff[r][c] = a[r][c] * gate[r / HW][c] + add[r / HW][c] * alpha
// (rows of one image share the gate; `add` is the pooled mean's gradient spread over the image) instead of a pass of its
// own writing a_eff out.
struct SeGate {
    const float* gate = nullptr;
    const float* add = nullptr;
    float alpha = 0.f;
    int HW = 1;
};

__device__ __forceinline__ float4 se_gated(float4 a, const SeGate& sg, unsigned row, unsigned c, unsigned C) {
    const size_t o = (size_t)(row / (unsigned)sg.HW) * C + c;
    const float4 g = *reinterpret_cast<const float4*>(sg.gate + o), d = *reinterpret_cast<const float4*>(sg.add + o);
    return make_float4(fmaf(a.x, g.x, d.x * sg.alpha), fmaf(a.y, g.y, d.y * sg.alpha), fmaf(a.z, g.z, d.z * sg.alpha), fmaf(a.w, g.w, d.w * sg.alpha));
}

template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ a, int lda, const float* __restrict__ x,
                                                          int ldx, const float* __restrict__ v1, const float* __restrict__ v2,
                                                          const float* __restrict__ v3, const float* __restrict__ v4,
                                                          const float* __restrict__ r1, const float* __restrict__ r2, int act,
                                                          int M, int C, float* __restrict__ part) {
    __shared__ float red[2][4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int chunk = blockIdx.y, nchunks = gridDim.y;
    const int r_lo = chunk * COL_ROWS, r_hi = min(M, r_lo + COL_ROWS);
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        float m = 0.f, rs = 0.f, sc = 0.f, bi = 0.f;
        if (MODE == COL_SQDEV || MODE == COL_BN_BWD || MODE == COL_MOMENTS) m = v1[c];
        if (MODE == COL_BN_BWD) { rs = v2[c]; sc = v3[c]; bi = v4[c]; }
        for (int r = r_lo + ry; r < r_hi; r += 4) {
            const float av = a[(size_t)r * lda + c];
            if (MODE == COL_SUM) {
                s1 += av;
            } else if (MODE == COL_SQDEV) {
                const float d = av - m;
                s1 += d * d;
            } else if (MODE == COL_BN_BWD) {
                const float xv = x[(size_t)r * ldx + c];
                const float g = av * act_grad(fmaf(xv, sc, bi), act);
                s1 += g;
                s2 += g * ((xv - m) * rs);
            } else if (MODE == COL_MOMENTS) {
                const float d = av - m;
                s1 += d;
                s2 += d * d;
            } else {
                const float xv = x[(size_t)r * ldx + c];
                s1 += av;
                s2 += av * ((xv - r1[r]) * r2[r]);
            }
        }
    }
    red[0][ry][cx] = s1;
    red[1][ry][cx] = s2;
    __syncthreads();
    if (ry == 0 && c < C) {
        part[(size_t)chunk * C + c] = ((red[0][0][cx] + red[0][1][cx]) + red[0][2][cx]) + red[0][3][cx];
        if (MODE >= COL_BN_BWD)
            part[(size_t)(nchunks + chunk) * C + c] = ((red[1][0][cx] + red[1][1][cx]) + red[1][2][cx]) + red[1][3][cx];
    }
}

// The same reduction for matrices whose rows can be read 16 bytes at a time (C % 4 == 0, row strides % 4 == 0):
// TX threads span 4*TX adjacent columns with one float4 each and 256/TX threads share the chunk's rows, so a
// wave reads whole contiguous rows whatever the channel count (16 ... 1152 in EfficientNet-B0); the row lanes
// meet in LDS in a fixed order.
// ACT >= 0 fixes the activation of the BatchNorm-gradient mode at compile time (a run-time switch per element costs a
// scalar branch tree around the transcendental instructions); ACT < 0 reads the `act` argument.
template <int MODE, int TX, int ACT = -1, bool GATED = false>
__global__ __launch_bounds__(256) void col_partial_vec_kernel(const float* __restrict__ a, int lda, const float* __restrict__ x,
                                                              int ldx, const float* __restrict__ v1, const float* __restrict__ v2,
                                                              const float* __restrict__ v3, const float* __restrict__ v4,
                                                              const float* __restrict__ r1, const float* __restrict__ r2, int act,
                                                              int M, int C, int rows_per_chunk, float* __restrict__ part, SeGate sg) {
    constexpr int TY = 256 / TX;
    __shared__ float4 red[2][TY][TX];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = (blockIdx.x * TX + tx) * 4;
    const int chunk = blockIdx.y, nchunks = gridDim.y;
    const int r_lo = chunk * rows_per_chunk, r_hi = min(M, r_lo + rows_per_chunk);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (c < C) {
        float4 m = s1, rs = s1, sc = s1, bi = s1;
        if (MODE == COL_SQDEV || MODE == COL_BN_BWD || MODE == COL_MOMENTS) m = *reinterpret_cast<const float4*>(v1 + c);
        if (MODE == COL_BN_BWD) {
            rs = *reinterpret_cast<const float4*>(v2 + c);
            sc = *reinterpret_cast<const float4*>(v3 + c);
            bi = *reinterpret_cast<const float4*>(v4 + c);
        }
        if (MODE == COL_BN_BWD) {
            // four rows per iteration, all eight loads issued before the arithmetic; rows beyond the chunk are read from a
            // clamped address and weighted 0
            const int actc = ACT >= 0 ? ACT : act;
            constexpr int RU = 4;
            for (int r = r_lo + ty; r < r_hi; r += RU * TY) {
                float4 av[RU], xv[RU];
                float wt[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int rr = min(r + u * TY, r_hi - 1);
                    wt[u] = r + u * TY < r_hi ? 1.f : 0.f;
                    av[u] = *reinterpret_cast<const float4*>(a + (size_t)rr * lda + c);
                    xv[u] = *reinterpret_cast<const float4*>(x + (size_t)rr * ldx + c);
                    if (GATED) av[u] = se_gated(av[u], sg, (unsigned)rr, (unsigned)c, (unsigned)C);
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const float g0 = wt[u] * av[u].x * act_grad(fmaf(xv[u].x, sc.x, bi.x), actc), g1 = wt[u] * av[u].y * act_grad(fmaf(xv[u].y, sc.y, bi.y), actc);
                    const float g2 = wt[u] * av[u].z * act_grad(fmaf(xv[u].z, sc.z, bi.z), actc), g3 = wt[u] * av[u].w * act_grad(fmaf(xv[u].w, sc.w, bi.w), actc);
                    s1.x += g0; s1.y += g1; s1.z += g2; s1.w += g3;
                    s2.x += g0 * ((xv[u].x - m.x) * rs.x); s2.y += g1 * ((xv[u].y - m.y) * rs.y);
                    s2.z += g2 * ((xv[u].z - m.z) * rs.z); s2.w += g3 * ((xv[u].w - m.w) * rs.w);
                }
            }
        } else
        for (int r = r_lo + ty; r < r_hi; r += TY) {
            const float4 av = *reinterpret_cast<const float4*>(a + (size_t)r * lda + c);
            if (MODE == COL_SUM) {
                s1.x += av.x; s1.y += av.y; s1.z += av.z; s1.w += av.w;
            } else if (MODE == COL_SQDEV) {
                const float dx = av.x - m.x, dy = av.y - m.y, dz = av.z - m.z, dw = av.w - m.w;
                s1.x += dx * dx; s1.y += dy * dy; s1.z += dz * dz; s1.w += dw * dw;
            } else if (MODE == COL_BN_BWD) {
                const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
                const float g0 = av.x * act_grad(fmaf(xv.x, sc.x, bi.x), act), g1 = av.y * act_grad(fmaf(xv.y, sc.y, bi.y), act);
                const float g2 = av.z * act_grad(fmaf(xv.z, sc.z, bi.z), act), g3 = av.w * act_grad(fmaf(xv.w, sc.w, bi.w), act);
                s1.x += g0; s1.y += g1; s1.z += g2; s1.w += g3;
                s2.x += g0 * ((xv.x - m.x) * rs.x); s2.y += g1 * ((xv.y - m.y) * rs.y);
                s2.z += g2 * ((xv.z - m.z) * rs.z); s2.w += g3 * ((xv.w - m.w) * rs.w);
            } else if (MODE == COL_MOMENTS) {
                const float dx = av.x - m.x, dy = av.y - m.y, dz = av.z - m.z, dw = av.w - m.w;
                s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
                s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
            } else {
                const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
                const float mr = r1[r], rr = r2[r];
                s1.x += av.x; s1.y += av.y; s1.z += av.z; s1.w += av.w;
                s2.x += av.x * ((xv.x - mr) * rr); s2.y += av.y * ((xv.y - mr) * rr);
                s2.z += av.z * ((xv.z - mr) * rr); s2.w += av.w * ((xv.w - mr) * rr);
            }
        }
    }
    red[0][ty][tx] = s1;
    red[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < C) {
        for (int pl = 0; pl < (MODE >= COL_BN_BWD ? 2 : 1); ++pl) {
            float4 t = red[pl][0][tx];
            for (int k = 1; k < TY; ++k) { const float4 o = red[pl][k][tx]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
            *reinterpret_cast<float4*>(part + (size_t)(pl * nchunks + chunk) * C + c) = t;
        }
    }
}

// out1[c] = scale * sum_chunks part[chunk][c] (+ out1[c] when accumulate); same for the second plane.
// 8 columns per workgroup, 32 lanes per column: lane ky adds chunks ky, ky + 32, ... - sixteen independent loads in
// flight at a time, so a 512-chunk column costs two memory round trips instead of a chain of 128 - and the 32 lanes
// meet in LDS; the order of additions is fixed.
__global__ __launch_bounds__(256) void col_finish_kernel(const float* __restrict__ part, int nchunks, int C, int planes,
                                                         float scale, int accumulate, float* __restrict__ out1,
                                                         float* __restrict__ out2) {
    __shared__ float red[32][8];
    const int cx = threadIdx.x & 7, ky = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cx;
    for (int pl = 0; pl < planes; ++pl) {
        float s = 0.f;
        if (c < C) {
            const float* src = part + (size_t)pl * nchunks * C + c;
            for (int k0 = ky; k0 < nchunks; k0 += 32 * 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k = k0 + 32 * u;
                    v[u] = k < nchunks ? src[(size_t)k * C] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) s += v[u];
            }
        }
        red[ky][cx] = s;
        __syncthreads();
        if (ky == 0 && c < C) {
            float t = red[0][cx];
#pragma unroll
            for (int k = 1; k < 32; ++k) t += red[k][cx];
            float* o = pl == 0 ? out1 : out2;
            o[c] = (accumulate ? o[c] : 0.f) + t * scale;
        }
        __syncthreads();
    }
}

// rows per chunk: a few hundred workgroups in flight for tall matrices, at least 64 rows each
// lanes along the channel axis (quads of channels) for the vector kernels: the width among 8 / 16 / 32 / 64 that leaves the
// fewest idle lanes in the last column block (the wider one on ties), e.g. 24 quads -> 8 (three full blocks, not 24 of 32)
static int col_vec_tx(int c4) {
    int best = 64, best_waste = 1 << 30;
    for (int tx = 64; tx >= 8; tx >>= 1) {
        const int waste = (c4 + tx - 1) / tx * tx - c4;
        if (waste < best_waste) { best_waste = waste; best = tx; }
    }
    return best;
}

// about 2048 workgroups in all (eight per CU: the reductions are bound by the bytes they keep in flight)
static int col_rows_per_chunk(int M, int C) {
    const int c4 = (C + 3) / 4, tx = col_vec_tx(c4);
    const int col_blocks = (c4 + tx - 1) / tx;
    long long want_chunks = 2048 / col_blocks;
    if (want_chunks < 1) want_chunks = 1;
    long long rows = (M + want_chunks - 1) / want_chunks;
    if (rows < 64) rows = 64;
    return (int)rows;
}

long long col_reduce_ws_floats(int M, int C) {
    const int rpc_vec = col_rows_per_chunk(M, C);
    const long long chunks_vec = (M + rpc_vec - 1) / rpc_vec, chunks_scalar = (M + COL_ROWS - 1) / COL_ROWS;
    return 2LL * (chunks_vec > chunks_scalar ? chunks_vec : chunks_scalar) * C;
}

template <int MODE, int ACT = -1, bool GATED = false>
static void col_launch_vec(int tx, dim3 grid, hipStream_t st, const float* a, int lda, const float* x, int ldx, const float* v1,
                           const float* v2, const float* v3, const float* v4, const float* r1, const float* r2, int act, int M, int C,
                           int rpc, float* ws, SeGate sg = SeGate{}) {
    switch (tx) {
        case 8: hipLaunchKernelGGL((col_partial_vec_kernel<MODE, 8, ACT, GATED>), grid, dim3(256), 0, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws, sg); break;
        case 16: hipLaunchKernelGGL((col_partial_vec_kernel<MODE, 16, ACT, GATED>), grid, dim3(256), 0, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws, sg); break;
        case 32: hipLaunchKernelGGL((col_partial_vec_kernel<MODE, 32, ACT, GATED>), grid, dim3(256), 0, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws, sg); break;
        default: hipLaunchKernelGGL((col_partial_vec_kernel<MODE, 64, ACT, GATED>), grid, dim3(256), 0, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws, sg); break;
    }
}

// first pass of a column reduction: per-chunk partial sums [planes][nchunks][C] into ws; returns the chunk count
static int col_partial_dispatch(int mode, const float* a, int lda, const float* x, int ldx, const float* v1, const float* v2,
                                const float* v3, const float* v4, const float* r1, const float* r2, int act, int M, int C, float* ws,
                                hipStream_t st);

int col_reduce_f32(int mode, const float* a, int lda, const float* x, int ldx, const float* v1, const float* v2,
                   const float* v3, const float* v4, const float* r1, const float* r2, int act, int M, int C, float scale,
                   int accumulate, float* out1, float* out2, float* ws, hipStream_t st) {
    if (!a || !out1 || !ws || M <= 0 || C <= 0 || lda < C || mode < 0 || mode > 4) { set_error("col_reduce: bad arguments"); return ICAP_ERR_ARG; }
    if ((mode == COL_SQDEV || mode == COL_MOMENTS) && !v1) { set_error("col_reduce: mean / shift vector required"); return ICAP_ERR_ARG; }
    if (mode == COL_MOMENTS && !out2) { set_error("col_reduce: two outputs required"); return ICAP_ERR_ARG; }
    if (mode == COL_BN_BWD && (!x || !v1 || !v2 || !v3 || !v4 || !out2 || ldx < C)) { set_error("col_reduce: BatchNorm operands required"); return ICAP_ERR_ARG; }
    if (mode == COL_LN_BWD && (!x || !r1 || !r2 || !out2 || ldx < C)) { set_error("col_reduce: LayerNorm operands required"); return ICAP_ERR_ARG; }
    const int nchunks = col_partial_dispatch(mode, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, ws, st);
    hipLaunchKernelGGL(col_finish_kernel, dim3((C + 7) / 8), dim3(256), 0, st, ws, nchunks, C, mode >= COL_BN_BWD ? 2 : 1, scale, accumulate, out1, out2);
    return launch_status("col_reduce");
}

static int col_partial_dispatch(int mode, const float* a, int lda, const float* x, int ldx, const float* v1, const float* v2,
                                const float* v3, const float* v4, const float* r1, const float* r2, int act, int M, int C, float* ws,
                                hipStream_t st) {
    const bool vec = (C % 4 == 0) && (lda % 4 == 0) && aligned16(a) && (!x || ((ldx % 4 == 0) && aligned16(x))) &&
                     (!v1 || aligned16(v1)) && (!v2 || aligned16(v2)) && (!v3 || aligned16(v3)) && (!v4 || aligned16(v4)) && aligned16(ws);
    int nchunks;
    if (vec) {
        const int c4 = C / 4;
        const int tx = col_vec_tx(c4);
        const int rpc = col_rows_per_chunk(M, C);
        nchunks = (M + rpc - 1) / rpc;
        const dim3 grid((c4 + tx - 1) / tx, nchunks);
        switch (mode) {
            case COL_SUM: col_launch_vec<COL_SUM>(tx, grid, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws); break;
            case COL_SQDEV: col_launch_vec<COL_SQDEV>(tx, grid, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws); break;
            case COL_BN_BWD:
                if (act == ACT_SWISH) col_launch_vec<COL_BN_BWD, ACT_SWISH>(tx, grid, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws);
                else if (act == ACT_NONE) col_launch_vec<COL_BN_BWD, ACT_NONE>(tx, grid, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws);
                else col_launch_vec<COL_BN_BWD>(tx, grid, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws);
                break;
            case COL_MOMENTS: col_launch_vec<COL_MOMENTS>(tx, grid, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws); break;
            default: col_launch_vec<COL_LN_BWD>(tx, grid, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, rpc, ws); break;
        }
    } else {
        nchunks = (M + COL_ROWS - 1) / COL_ROWS;
        const dim3 grid((C + 63) / 64, nchunks);
        switch (mode) {
            case COL_SUM: hipLaunchKernelGGL(col_partial_kernel<COL_SUM>, grid, dim3(256), 0, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, ws); break;
            case COL_SQDEV: hipLaunchKernelGGL(col_partial_kernel<COL_SQDEV>, grid, dim3(256), 0, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, ws); break;
            case COL_BN_BWD: hipLaunchKernelGGL(col_partial_kernel<COL_BN_BWD>, grid, dim3(256), 0, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, ws); break;
            case COL_MOMENTS: hipLaunchKernelGGL(col_partial_kernel<COL_MOMENTS>, grid, dim3(256), 0, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, ws); break;
            default: hipLaunchKernelGGL(col_partial_kernel<COL_LN_BWD>, grid, dim3(256), 0, st, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, ws); break;
        }
    }
    return nchunks;
}

// ---------------------------------------------------------------- BatchNorm (training mode)
// From the batch mean and the summed squared deviations: rstd, the fused scale / bias of the
// normalisation (y = x*s + b) and the running statistics update
//   running = (1 - momentum) * running + momentum * batch   (variance: unbiased, n / (n - 1)).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ mean, const float* __restrict__ sqdev,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, int M, int C, float* __restrict__ rstd,
                                                          float* __restrict__ s, float* __restrict__ b,
                                                          float* __restrict__ run_mean, float* __restrict__ run_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float var = sqdev[c] / (float)M;
    const float r = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f;
    rstd[c] = r;
    s[c] = g * r;
    b[c] = (beta ? beta[c] : 0.f) - mean[c] * g * r;
    if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean[c];
    if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * (M > 1 ? sqdev[c] / (float)(M - 1) : var);
}

// The same from one-pass shifted moments: s1 = sum (x - shift), s2 = sum (x - shift)^2 over the M rows.
//   mean = shift + s1 / M,  sum of squared deviations = s2 - s1^2 / M  (the shift - any row of x - keeps the subtraction benign)
__global__ __launch_bounds__(256) void bn_finalize_moments_kernel(const float* __restrict__ shift, const float* __restrict__ s1,
                                                                  const float* __restrict__ s2, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps, float momentum, int M,
                                                                  int C, float* __restrict__ mean, float* __restrict__ rstd,
                                                                  float* __restrict__ s, float* __restrict__ b,
                                                                  float* __restrict__ run_mean, float* __restrict__ run_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float a1 = s1[c];
    const float mu = shift[c] + a1 / (float)M;
    const float sq = fmaxf(s2[c] - a1 * a1 / (float)M, 0.f);
    const float var = sq / (float)M;
    const float r = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f;
    mean[c] = mu;
    rstd[c] = r;
    s[c] = g * r;
    b[c] = (beta ? beta[c] : 0.f) - mu * g * r;
    if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
    if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * (M > 1 ? sq / (float)(M - 1) : var);
}

int bn_finalize_moments_f32(const float* shift, const float* s1, const float* s2, const float* gamma, const float* beta, float eps,
                            float momentum, int M, int C, float* mean, float* rstd, float* s, float* b, float* run_mean,
                            float* run_var, hipStream_t st) {
    if (!shift || !s1 || !s2 || !mean || !rstd || !s || !b || M <= 0 || C <= 0) { set_error("bn_finalize_moments: bad arguments"); return ICAP_ERR_ARG; }
    hipLaunchKernelGGL(bn_finalize_moments_kernel, dim3((C + 255) / 256), dim3(256), 0, st, shift, s1, s2, gamma, beta, eps, momentum, M, C, mean, rstd, s, b, run_mean, run_var);
    return launch_status("bn_finalize_moments");
}

// second pass of the one-pass statistics, fused with the finalisation: the chunk sums of both moment planes are added as in
// col_finish_kernel (fixed order), then mean / rstd / fused scale and bias / running statistics are written - one launch less
// per BatchNorm than col_reduce + bn_finalize_moments
__global__ __launch_bounds__(256) void col_finish_bn_kernel(const float* __restrict__ part, int nchunks, const float* __restrict__ shift,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float momentum, int M, int C, float* __restrict__ mean,
                                                            float* __restrict__ rstd, float* __restrict__ s, float* __restrict__ b,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var) {
    __shared__ float red[32][8];
    const int cx = threadIdx.x & 7, ky = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cx;
    float tot[2] = {0.f, 0.f};
    for (int pl = 0; pl < 2; ++pl) {
        float acc = 0.f;
        if (c < C) {
            const float* src = part + (size_t)pl * nchunks * C + c;
            for (int k0 = ky; k0 < nchunks; k0 += 32 * 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int k = k0 + 32 * u;
                    v[u] = k < nchunks ? src[(size_t)k * C] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += v[u];
            }
        }
        red[ky][cx] = acc;
        __syncthreads();
        if (ky == 0) {
            float t = red[0][cx];
#pragma unroll
            for (int k = 1; k < 32; ++k) t += red[k][cx];
            tot[pl] = t;
        }
        __syncthreads();
    }
    if (ky != 0 || c >= C) return;
    const float a1 = tot[0];
    const float mu = shift[c] + a1 / (float)M;
    const float sq = fmaxf(tot[1] - a1 * a1 / (float)M, 0.f);
    const float var = sq / (float)M;
    const float r = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f;
    mean[c] = mu;
    rstd[c] = r;
    s[c] = g * r;
    b[c] = (beta ? beta[c] : 0.f) - mu * g * r;
    if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
    if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * (M > 1 ? sq / (float)(M - 1) : var);
}

// batch statistics of x [M][C] in one pass over x (moments about its first row) and everything a training-mode BatchNorm
// derives from them; ws as for col_reduce_f32
int bn_train_stats_f32(const float* x, int ldx, const float* gamma, const float* beta, float eps, float momentum, int M, int C,
                       float* mean, float* rstd, float* s, float* b, float* run_mean, float* run_var, float* ws, hipStream_t st) {
    if (!x || !mean || !rstd || !s || !b || !ws || M <= 0 || C <= 0 || ldx < C) { set_error("bn_train_stats: bad arguments"); return ICAP_ERR_ARG; }
    const int nchunks = col_partial_dispatch(COL_MOMENTS, x, ldx, nullptr, 0, x, nullptr, nullptr, nullptr, nullptr, nullptr, ACT_NONE, M, C, ws, st);
    hipLaunchKernelGGL(col_finish_bn_kernel, dim3((C + 7) / 8), dim3(256), 0, st, ws, nchunks, x, gamma, beta, eps, momentum, M, C, mean,
                       rstd, s, b, run_mean, run_var);
    return launch_status("bn_train_stats");
}

// ---- train-mode BatchNorm statistics from the partials the PRODUCING kernel left behind (the GEMM epilogues'
// tile_col_stats, the depthwise kernel's per-workgroup sums): part [P][2][C] holds, per block of rows, every column's sum
// and its squared deviations about the block's own mean; cnt [P] the block's row count (null: block p covers
// min(rows_per_part, M - p * rows_per_part) rows).  Blocks are merged pairwise with Chan's update
//   n = na + nb,  d = sb / nb - sa / na,  q = qa + qb + d^2 na nb / n
// in a fixed order (a thread walks its partials in index order, the four row slots of a workgroup and then the chunks are
// merged in index order), so the result does not depend on scheduling and no squares of un-centred values are ever formed.
constexpr int STAT_U = 8;   // partials a thread keeps in flight

// A workgroup = TX columns x 256 / TX row slots.  Slot k walks entries e0 + k, e0 + k + slots, ... of [e0, e1) with STAT_U
// loads in flight, then the slots are merged in order; slot 0 returns the column's statistics.  TRIPLE: entries are
// (n, sum, q) triples [k][3][C] (the chunk results); otherwise raw partials [p][2][C] with their row counts.
template <bool TRIPLE, int TX>
__device__ __forceinline__ ColStat stat_walk(const float* __restrict__ src, const float* __restrict__ cnt, int e0, int e1,
                                             int rows_per_part, int M, int C, int c, float (*red)[3][TX]) {
    constexpr int TY = 256 / TX;
    const int cx = threadIdx.x % TX, slot = threadIdx.x / TX;
    ColStat a = {0.f, 0.f, 0.f};
    if (c < C) {
        for (int eb = e0 + slot; eb < e1; eb += TY * STAT_U) {
            float sv[STAT_U], qv[STAT_U], nv[STAT_U];
#pragma unroll
            for (int u = 0; u < STAT_U; ++u) {
                const int ee = eb + TY * u;
                const int ec = min(ee, e1 - 1);
                float n_here;
                if constexpr (TRIPLE) {
                    n_here = src[((size_t)ec * 3 + 0) * C + c];
                    sv[u] = src[((size_t)ec * 3 + 1) * C + c];
                    qv[u] = src[((size_t)ec * 3 + 2) * C + c];
                } else {
                    sv[u] = src[((size_t)ec * 2 + 0) * C + c];
                    qv[u] = src[((size_t)ec * 2 + 1) * C + c];
                    n_here = cnt ? cnt[ec] : (float)min(rows_per_part, max(M - ec * rows_per_part, 0));
                }
                nv[u] = ee < e1 ? n_here : 0.f;
            }
#pragma unroll
            for (int u = 0; u < STAT_U; ++u) stat_merge(a, nv[u], sv[u], qv[u]);
        }
    }
    red[slot][0][cx] = a.n; red[slot][1][cx] = a.s; red[slot][2][cx] = a.q;
    __syncthreads();
    if (slot == 0) {
#pragma unroll
        for (int k = 1; k < TY; ++k) stat_merge(a, red[k][0][cx], red[k][1][cx], red[k][2][cx]);
    }
    return a;
}

// level 1: workgroup (column block of 64, chunk of ppc partials) -> chunk_out [nchunk][3][C] = (n, sum, q)
__global__ __launch_bounds__(256) void stat_chunk_kernel(const float* __restrict__ part, const float* __restrict__ cnt, int P, int ppc,
                                                         int rows_per_part, int M, int C, float* __restrict__ chunk_out) {
    __shared__ float red[4][3][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int p0 = blockIdx.y * ppc, p1 = min(P, p0 + ppc);
    const ColStat a = stat_walk<false, 64>(part, cnt, p0, p1, rows_per_part, M, C, c, red);
    if ((threadIdx.x >> 6) == 0 && c < C) {
        float* o = chunk_out + (size_t)blockIdx.y * 3 * C;
        o[c] = a.n; o[C + c] = a.s; o[2 * C + c] = a.q;
    }
}

// level 2: merge the chunks (TRIPLE) - or, for a few hundred partials, the partials themselves in one launch - and derive
// what the BatchNorm forward / backward need; updates the running statistics
template <bool TRIPLE>
__global__ __launch_bounds__(256) void stat_finish_bn_kernel(const float* __restrict__ src, const float* __restrict__ cnt, int nent,
                                                             int rows_per_part, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, float momentum, int M, int C,
                                                             float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ s,
                                                             float* __restrict__ b, float* __restrict__ run_mean, float* __restrict__ run_var) {
    // 16 columns x 16 slots: a few hundred entries per column are two to four rounds of loads
    __shared__ float red[16][3][16];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const ColStat a = stat_walk<TRIPLE, 16>(src, cnt, 0, nent, rows_per_part, M, C, c, red);
    if ((threadIdx.x >> 4) != 0 || c >= C) return;
    const float mu = a.n > 0.f ? a.s / a.n : 0.f;
    const float sq = fmaxf(a.q, 0.f);
    const float var = sq / (float)M;
    const float r = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f;
    mean[c] = mu;
    rstd[c] = r;
    s[c] = g * r;
    b[c] = (beta ? beta[c] : 0.f) - mu * g * r;
    if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
    if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * (M > 1 ? sq / (float)(M - 1) : var);
}

// chunks of 32 partials (one round of loads per thread), doubled until at most 256 chunks are left for the second level
constexpr int STAT_ONE_LAUNCH = 512;   // up to this many partials the finishing kernel reads them itself

static void stat_geometry(int P, int* nchunk, int* ppc) {
    int per = 32;
    while ((P + per - 1) / per > 256) per *= 2;
    *ppc = per;
    *nchunk = (P + per - 1) / per;
}

long long bn_stats_from_parts_ws_floats(int P, int C) {
    if (P <= 0 || C <= 0) return 0;
    int nc, ppc;
    stat_geometry(P, &nc, &ppc);
    return 3LL * nc * C;
}

int bn_stats_from_parts_f32(const float* part, const float* cnt, int P, int rows_per_part, int M, int C, const float* gamma,
                            const float* beta, float eps, float momentum, float* mean, float* rstd, float* s, float* b,
                            float* run_mean, float* run_var, float* ws, hipStream_t st) {
    if (!part || !mean || !rstd || !s || !b || !ws || P <= 0 || M <= 0 || C <= 0 || (!cnt && rows_per_part <= 0)) {
        set_error("bn_stats_from_parts: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (!cnt && (long long)P * rows_per_part < M) {
        set_error("bn_stats_from_parts: %d partials of %d rows do not cover %d rows", P, rows_per_part, M);
        return ICAP_ERR_ARG;
    }
    if (P <= STAT_ONE_LAUNCH) {
        hipLaunchKernelGGL(stat_finish_bn_kernel<false>, dim3((C + 15) / 16), dim3(256), 0, st, part, cnt, P, rows_per_part, gamma, beta, eps,
                           momentum, M, C, mean, rstd, s, b, run_mean, run_var);
        return launch_status("bn_stats_from_parts");
    }
    int nc, ppc;
    stat_geometry(P, &nc, &ppc);
    hipLaunchKernelGGL(stat_chunk_kernel, dim3((C + 63) / 64, nc), dim3(256), 0, st, part, cnt, P, ppc, rows_per_part, M, C, ws);
    hipLaunchKernelGGL(stat_finish_bn_kernel<true>, dim3((C + 15) / 16), dim3(256), 0, st, ws, nullptr, nc, 0, gamma, beta, eps, momentum, M, C,
                       mean, rstd, s, b, run_mean, run_var);
    return launch_status("bn_stats_from_parts");
}

int bn_finalize_f32(const float* mean, const float* sqdev, const float* gamma, const float* beta, float eps, float momentum,
                    int M, int C, float* rstd, float* s, float* b, float* run_mean, float* run_var, hipStream_t st) {
    if (!mean || !sqdev || !rstd || !s || !b || M <= 0 || C <= 0) { set_error("bn_finalize: bad arguments"); return ICAP_ERR_ARG; }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, mean, sqdev, gamma, beta, eps, momentum, M, C, rstd, s, b, run_mean, run_var);
    return launch_status("bn_finalize");
}

// y[r][c] = act(x[r][c] * s[c] + b[c]) (+ res[r][c]); s, b may be null (1, 0)
__global__ __launch_bounds__(256) void scale_bias_act_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ s,
                                                             const float* __restrict__ b, const float* __restrict__ res,
                                                             int ldr, int act, int M, int C, float* __restrict__ y, int ldy) {
    const long long total = (long long)M * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / C), c = (int)(i - (long long)r * C);
        float v = act_apply(fmaf(x[(size_t)r * ldx + c], s ? s[c] : 1.f, b ? b[c] : 0.f), act);
        if (res) v += res[(size_t)r * ldr + c];
        y[(size_t)r * ldy + c] = v;
    }
}

// 16 bytes per lane when every row start and the channel count allow it
template <int ACT>      // >= 0: the activation fixed at compile time; < 0: the `act` argument decides per element
__global__ __launch_bounds__(256) void scale_bias_act_vec_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ s,
                                                                 const float* __restrict__ b, const float* __restrict__ res,
                                                                 int ldr, int act, int M, int C4, float* __restrict__ y, int ldy) {
    const unsigned total = (unsigned)M * (unsigned)C4;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned r = i / (unsigned)C4, c = (i - r * (unsigned)C4) << 2;
        const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
        const float4 sv = s ? *reinterpret_cast<const float4*>(s + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 bv = b ? *reinterpret_cast<const float4*>(b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const int a_ = ACT >= 0 ? ACT : act;
        float4 v = make_float4(act_apply(fmaf(xv.x, sv.x, bv.x), a_), act_apply(fmaf(xv.y, sv.y, bv.y), a_),
                               act_apply(fmaf(xv.z, sv.z, bv.z), a_), act_apply(fmaf(xv.w, sv.w, bv.w), a_));
        if (res) {
            const float4 rv = *reinterpret_cast<const float4*>(res + (size_t)r * ldr + c);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        *reinterpret_cast<float4*>(y + (size_t)r * ldy + c) = v;
    }
}

static bool vec4_ok(long long M, int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
    if (C % 4 || M * (C / 4) >= (1LL << 31)) return false;
    for (int ld : lds) if (ld % 4) return false;
    for (const void* p : ptrs) if (p && !aligned16(p)) return false;
    return true;
}

int scale_bias_act_f32(const float* x, int ldx, const float* s, const float* b, const float* res, int ldr, int act, int M, int C,
                       float* y, int ldy, hipStream_t st) {
    if (!x || !y || M <= 0 || C <= 0 || ldx < C || ldy < C || (res && ldr < C)) { set_error("scale_bias_act: bad arguments"); return ICAP_ERR_ARG; }
    if (vec4_ok(M, C, {ldx, ldy, res ? ldr : 0}, {x, y, s, b, res})) {
        long long vb = ((long long)M * (C / 4) + 255) / 256;
        if (vb > 16384) vb = 16384;
        if (act == ACT_SWISH)
            hipLaunchKernelGGL(scale_bias_act_vec_kernel<ACT_SWISH>, dim3((unsigned)vb), dim3(256), 0, st, x, ldx, s, b, res, ldr, act, M, C / 4, y, ldy);
        else if (act == ACT_NONE)
            hipLaunchKernelGGL(scale_bias_act_vec_kernel<ACT_NONE>, dim3((unsigned)vb), dim3(256), 0, st, x, ldx, s, b, res, ldr, act, M, C / 4, y, ldy);
        else
            hipLaunchKernelGGL(scale_bias_act_vec_kernel<-1>, dim3((unsigned)vb), dim3(256), 0, st, x, ldx, s, b, res, ldr, act, M, C / 4, y, ldy);
        return launch_status("scale_bias_act");
    }
    long long blocks = ((long long)M * C + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_bias_act_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx, s, b, res, ldr, act, M, C, y, ldy);
    return launch_status("scale_bias_act");
}

// BatchNorm + activation, gradient with respect to the input:
//   g = dy * act'(x*s + b);  dx = s[c] * (g - dbeta[c]/M - xhat * dgamma[c]/M),  xhat = (x - mean[c]) * rstd[c]
// With mean == null the layer is a plain per-channel affine + activation (eval-mode statistics): dx = s[c] * g.
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float* __restrict__ dy, int ldd, const float* __restrict__ x, int ldx,
                                                         const float* __restrict__ s, const float* __restrict__ b,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                         int act, int M, int C, float* __restrict__ dx, int ldo) {
    const long long total = (long long)M * C;
    const float inv_m = 1.0f / (float)M;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / C), c = (int)(i - (long long)r * C);
        const float xv = x[(size_t)r * ldx + c];
        const float sc = s ? s[c] : 1.f;
        float g = dy[(size_t)r * ldd + c] * act_grad(fmaf(xv, sc, b ? b[c] : 0.f), act);
        if (mean) g = g - dbeta[c] * inv_m - (xv - mean[c]) * rstd[c] * dgamma[c] * inv_m;
        dx[(size_t)r * ldo + c] = sc * g;
    }
}

// The vector form: a thread owns ONE channel quad for the whole launch (the grid's stride is a multiple of the quads per
// row), so the per-channel parameters are loaded once, and it keeps BWD_U row positions - 2 x BWD_U 16-byte loads - in flight.
constexpr int BWD_U = 4;

template <int ACT, bool GATED = false>      // ACT >= 0: the activation fixed at compile time; < 0: the `act` argument decides per element
__global__ __launch_bounds__(256) void bn_act_bwd_vec_kernel(const float* __restrict__ dy, int ldd, const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ s, const float* __restrict__ b,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                             int act, int M, int C4, float* __restrict__ dx, int ldo, SeGate sg) {
    const float inv_m = 1.0f / (float)M;
    const unsigned gtid = blockIdx.x * 256u + threadIdx.x;
    const unsigned stride = gridDim.x * 256u;                 // a multiple of C4 (the launcher sees to it)
    const unsigned row_step = stride / (unsigned)C4;
    unsigned r = gtid / (unsigned)C4;
    const unsigned c = (gtid - r * (unsigned)C4) << 2;
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 s4 = s ? *reinterpret_cast<const float4*>(s + c) : one;
    const float4 b4 = b ? *reinterpret_cast<const float4*>(b + c) : zero;
    const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
    float mv[4] = {0.f, 0.f, 0.f, 0.f}, kv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {0.f, 0.f, 0.f, 0.f}, ev[4] = {0.f, 0.f, 0.f, 0.f};
    if (mean) {
        const float4 m4 = *reinterpret_cast<const float4*>(mean + c), r4 = *reinterpret_cast<const float4*>(rstd + c);
        const float4 g4 = *reinterpret_cast<const float4*>(dgamma + c), e4 = *reinterpret_cast<const float4*>(dbeta + c);
        mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
        kv[0] = r4.x; kv[1] = r4.y; kv[2] = r4.z; kv[3] = r4.w;
        gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
        ev[0] = e4.x * inv_m; ev[1] = e4.y * inv_m; ev[2] = e4.z * inv_m; ev[3] = e4.w * inv_m;
    }
    for (; r < (unsigned)M; r += row_step * BWD_U) {
        float4 xq[BWD_U], dq[BWD_U];
#pragma unroll
        for (int u = 0; u < BWD_U; ++u) {
            const unsigned rr = min(r + u * row_step, (unsigned)M - 1u);     // clamped: every load is issued before the first use
            xq[u] = *reinterpret_cast<const float4*>(x + (size_t)rr * ldx + c);
            dq[u] = *reinterpret_cast<const float4*>(dy + (size_t)rr * ldd + c);
            if (GATED) dq[u] = se_gated(dq[u], sg, rr, c, (unsigned)C4 * 4u);
        }
#pragma unroll
        for (int u = 0; u < BWD_U; ++u) {
            const unsigned rr = r + u * row_step;
            if (rr >= (unsigned)M) break;
            const float xv[4] = {xq[u].x, xq[u].y, xq[u].z, xq[u].w}, dv[4] = {dq[u].x, dq[u].y, dq[u].z, dq[u].w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = dv[k] * act_grad(fmaf(xv[k], sv[k], bv[k]), ACT >= 0 ? ACT : act);
            if (mean) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = o[k] - ev[k] - (xv[k] - mv[k]) * kv[k] * gv[k] * inv_m;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] *= sv[k];
            *reinterpret_cast<float4*>(dx + (size_t)rr * ldo + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

int bn_act_bwd_f32(const float* dy, int ldd, const float* x, int ldx, const float* s, const float* b, const float* mean,
                   const float* rstd, const float* dgamma, const float* dbeta, int act, int M, int C, float* dx, int ldo,
                   hipStream_t st) {
    if (!dy || !x || !dx || M <= 0 || C <= 0 || ldd < C || ldx < C || ldo < C || (mean && (!rstd || !dgamma || !dbeta))) {
        set_error("bn_act_bwd: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (vec4_ok(M, C, {ldd, ldx, ldo}, {dy, x, dx, s, b, mean, rstd, dgamma, dbeta})) {
        // grid x 256 threads must be a multiple of the quads per row (a thread keeps its channel quad): a multiple of
        // C4 / gcd(C4, 256) workgroups, enough of them that a thread walks about BWD_U x 4 rows
        const int C4 = C / 4;
        int gcd = C4, t256 = 256;
        while (t256) { const int t = gcd % t256; gcd = t256; t256 = t; }
        const long long mult = C4 / gcd;
        long long vb = ((long long)M * C4 + 256LL * BWD_U * 4 - 1) / (256LL * BWD_U * 4);
        if (vb > 16384) vb = 16384;
        vb = (vb + mult - 1) / mult * mult;
        if (act == ACT_SWISH)
            hipLaunchKernelGGL(bn_act_bwd_vec_kernel<ACT_SWISH>, dim3((unsigned)vb), dim3(256), 0, st, dy, ldd, x, ldx, s, b, mean, rstd, dgamma,
                               dbeta, act, M, C / 4, dx, ldo, SeGate{});
        else if (act == ACT_NONE)
            hipLaunchKernelGGL(bn_act_bwd_vec_kernel<ACT_NONE>, dim3((unsigned)vb), dim3(256), 0, st, dy, ldd, x, ldx, s, b, mean, rstd, dgamma,
                               dbeta, act, M, C / 4, dx, ldo, SeGate{});
        else
            hipLaunchKernelGGL(bn_act_bwd_vec_kernel<-1>, dim3((unsigned)vb), dim3(256), 0, st, dy, ldd, x, ldx, s, b, mean, rstd, dgamma, dbeta,
                               act, M, C / 4, dx, ldo, SeGate{});
        return launch_status("bn_act_bwd");
    }
    long long blocks = ((long long)M * C + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, ldd, x, ldx, s, b, mean, rstd, dgamma, dbeta, act, M, C, dx, ldo);
    return launch_status("bn_act_bwd");
}

// ---------------------------------------------------------------- BatchNorm1d of every decode step, all steps per launch
// The teacher-forced Bahdanau decoder normalises the V logits of each time step with the statistics of that step's B rows.
// x [T*B][V] is step-major (rows t*B ... t*B + B - 1 belong to step t); y / dy are [B][T][V], the layout the loss reads.
// One workgroup owns 256 columns of one step: 64 channel-quad lanes x 4 row lanes, four rows in flight per thread, the row
// lanes meet in LDS in a fixed order.  Per-step launches of the plain kernels are bound by launch latency at these sizes
// (6.9 MB per step at B = 256, V = 6756).
constexpr int BNS_TX = 64, BNS_TY = 4, BNS_RU = 4;

__global__ __launch_bounds__(256) void bn_steps_stats_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int B, int V,
                                                             float* __restrict__ mean, float* __restrict__ rstd,
                                                             float* __restrict__ s, float* __restrict__ b, float* __restrict__ uvar) {
    __shared__ float4 red[2][BNS_TY][BNS_TX];
    const int tx = threadIdx.x % BNS_TX, ty = threadIdx.x / BNS_TX;
    const int c = (blockIdx.x * BNS_TX + tx) * 4, t = blockIdx.y;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, sh = s1;
    if (c < V) {
        const float* base = x + (size_t)t * B * V + c;
        sh = *reinterpret_cast<const float4*>(base);                   // moments about the step's first row
        for (int r = ty; r < B; r += BNS_TY * BNS_RU) {
            float4 v[BNS_RU];
#pragma unroll
            for (int u = 0; u < BNS_RU; ++u) v[u] = *reinterpret_cast<const float4*>(base + (size_t)min(r + u * BNS_TY, B - 1) * V);
#pragma unroll
            for (int u = 0; u < BNS_RU; ++u) {
                if (r + u * BNS_TY >= B) break;
                const float d0 = v[u].x - sh.x, d1 = v[u].y - sh.y, d2 = v[u].z - sh.z, d3 = v[u].w - sh.w;
                s1.x += d0; s1.y += d1; s1.z += d2; s1.w += d3;
                s2.x = fmaf(d0, d0, s2.x); s2.y = fmaf(d1, d1, s2.y); s2.z = fmaf(d2, d2, s2.z); s2.w = fmaf(d3, d3, s2.w);
            }
        }
    }
    red[0][ty][tx] = s1;
    red[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < V) {
        float a1[4], a2[4];
        {
            float4 p = red[0][0][tx], q = red[1][0][tx];
            for (int k = 1; k < BNS_TY; ++k) {
                const float4 o = red[0][k][tx], w = red[1][k][tx];
                p.x += o.x; p.y += o.y; p.z += o.z; p.w += o.w;
                q.x += w.x; q.y += w.y; q.z += w.z; q.w += w.w;
            }
            a1[0] = p.x; a1[1] = p.y; a1[2] = p.z; a1[3] = p.w;
            a2[0] = q.x; a2[1] = q.y; a2[2] = q.z; a2[3] = q.w;
        }
        const float shv[4] = {sh.x, sh.y, sh.z, sh.w};
        const float4 g4 = gamma ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 b4 = beta ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
        float mu[4], rs[4], sc[4], bi[4], uv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mu[k] = shv[k] + a1[k] / (float)B;
            const float sq = fmaxf(a2[k] - a1[k] * a1[k] / (float)B, 0.f);
            const float var = sq / (float)B;
            rs[k] = 1.0f / sqrtf(var + eps);
            sc[k] = gv[k] * rs[k];
            bi[k] = bv[k] - mu[k] * gv[k] * rs[k];
            uv[k] = B > 1 ? sq / (float)(B - 1) : var;
        }
        const size_t o = (size_t)t * V + c;
        *reinterpret_cast<float4*>(mean + o) = make_float4(mu[0], mu[1], mu[2], mu[3]);
        *reinterpret_cast<float4*>(rstd + o) = make_float4(rs[0], rs[1], rs[2], rs[3]);
        *reinterpret_cast<float4*>(s + o) = make_float4(sc[0], sc[1], sc[2], sc[3]);
        *reinterpret_cast<float4*>(b + o) = make_float4(bi[0], bi[1], bi[2], bi[3]);
        *reinterpret_cast<float4*>(uvar + o) = make_float4(uv[0], uv[1], uv[2], uv[3]);
    }
}

// running statistics: one momentum update per step, in step order
__global__ __launch_bounds__(256) void bn_steps_running_kernel(const float* __restrict__ mean, const float* __restrict__ uvar,
                                                               float momentum, int T, int V, float* __restrict__ run_mean,
                                                               float* __restrict__ run_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= V) return;
    float rm = run_mean ? run_mean[c] : 0.f, rv = run_var ? run_var[c] : 0.f;
    for (int t = 0; t < T; ++t) {
        rm = (1.f - momentum) * rm + momentum * mean[(size_t)t * V + c];
        rv = (1.f - momentum) * rv + momentum * uvar[(size_t)t * V + c];
    }
    if (run_mean) run_mean[c] = rm;
    if (run_var) run_var[c] = rv;
}

// y[r][t][:] = x[t*B + r][:] * s[t][:] + b[t][:]
__global__ __launch_bounds__(256) void bn_steps_apply_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                             const float* __restrict__ b, int B, int T, int V4,
                                                             float* __restrict__ y) {
    const unsigned total = (unsigned)T * (unsigned)B * (unsigned)V4;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned row = i / (unsigned)V4, c4 = i - row * (unsigned)V4;
        const unsigned t = row / (unsigned)B, r = row - t * (unsigned)B;
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        const float4 sv = reinterpret_cast<const float4*>(s)[(size_t)t * V4 + c4], bv = reinterpret_cast<const float4*>(b)[(size_t)t * V4 + c4];
        reinterpret_cast<float4*>(y)[((size_t)r * T + t) * V4 + c4] =
            make_float4(fmaf(xv.x, sv.x, bv.x), fmaf(xv.y, sv.y, bv.y), fmaf(xv.z, sv.z, bv.z), fmaf(xv.w, sv.w, bv.w));
    }
}

// dbeta_t[t][c] = sum_r dy[r][t][c],  dgamma_t[t][c] = sum_r dy[r][t][c] * xhat[t*B + r][c]
__global__ __launch_bounds__(256) void bn_steps_bwd_sums_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                int B, int T, int V, float* __restrict__ dgamma_t,
                                                                float* __restrict__ dbeta_t) {
    __shared__ float4 red[2][BNS_TY][BNS_TX];
    const int tx = threadIdx.x % BNS_TX, ty = threadIdx.x / BNS_TX;
    const int c = (blockIdx.x * BNS_TX + tx) * 4, t = blockIdx.y;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    if (c < V) {
        const float* xb = x + (size_t)t * B * V + c;
        const float* db = dy + (size_t)t * V + c;
        const float4 m = *reinterpret_cast<const float4*>(mean + (size_t)t * V + c), rs = *reinterpret_cast<const float4*>(rstd + (size_t)t * V + c);
        for (int r = ty; r < B; r += BNS_TY * BNS_RU) {
            float4 xv[BNS_RU], dv[BNS_RU];
#pragma unroll
            for (int u = 0; u < BNS_RU; ++u) {
                const int rr = min(r + u * BNS_TY, B - 1);
                xv[u] = *reinterpret_cast<const float4*>(xb + (size_t)rr * V);
                dv[u] = *reinterpret_cast<const float4*>(db + (size_t)rr * T * V);
            }
#pragma unroll
            for (int u = 0; u < BNS_RU; ++u) {
                if (r + u * BNS_TY >= B) break;
                s1.x += dv[u].x; s1.y += dv[u].y; s1.z += dv[u].z; s1.w += dv[u].w;
                s2.x += dv[u].x * ((xv[u].x - m.x) * rs.x); s2.y += dv[u].y * ((xv[u].y - m.y) * rs.y);
                s2.z += dv[u].z * ((xv[u].z - m.z) * rs.z); s2.w += dv[u].w * ((xv[u].w - m.w) * rs.w);
            }
        }
    }
    red[0][ty][tx] = s1;
    red[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && c < V) {
        float4 p = red[0][0][tx], q = red[1][0][tx];
        for (int k = 1; k < BNS_TY; ++k) {
            const float4 o = red[0][k][tx], w = red[1][k][tx];
            p.x += o.x; p.y += o.y; p.z += o.z; p.w += o.w;
            q.x += w.x; q.y += w.y; q.z += w.z; q.w += w.w;
        }
        *reinterpret_cast<float4*>(dbeta_t + (size_t)t * V + c) = p;
        *reinterpret_cast<float4*>(dgamma_t + (size_t)t * V + c) = q;
    }
}

// dx[t*B + r][c] = s[t][c] * (dy[r][t][c] - dbeta_t[t][c] / B - xhat * dgamma_t[t][c] / B)
__global__ __launch_bounds__(256) void bn_steps_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ s, const float* __restrict__ dgamma_t,
                                                                 const float* __restrict__ dbeta_t, int B, int T, int V4,
                                                                 float* __restrict__ dx) {
    const unsigned total = (unsigned)T * (unsigned)B * (unsigned)V4;
    const float inv_b = 1.0f / (float)B;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned row = i / (unsigned)V4, c4 = i - row * (unsigned)V4;
        const unsigned t = row / (unsigned)B, r = row - t * (unsigned)B;
        const size_t o = (size_t)t * V4 + c4;
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        const float4 dv = reinterpret_cast<const float4*>(dy)[((size_t)r * T + t) * V4 + c4];
        const float4 m = reinterpret_cast<const float4*>(mean)[o], rs = reinterpret_cast<const float4*>(rstd)[o];
        const float4 sv = reinterpret_cast<const float4*>(s)[o];
        const float4 dg = reinterpret_cast<const float4*>(dgamma_t)[o], db = reinterpret_cast<const float4*>(dbeta_t)[o];
        reinterpret_cast<float4*>(dx)[i] = make_float4(sv.x * (dv.x - db.x * inv_b - (xv.x - m.x) * rs.x * dg.x * inv_b),
                                                       sv.y * (dv.y - db.y * inv_b - (xv.y - m.y) * rs.y * dg.y * inv_b),
                                                       sv.z * (dv.z - db.z * inv_b - (xv.z - m.z) * rs.z * dg.z * inv_b),
                                                       sv.w * (dv.w - db.w * inv_b - (xv.w - m.w) * rs.w * dg.w * inv_b));
    }
}

static bool bn_steps_ok(int B, int T, int V, std::initializer_list<const void*> ptrs) {
    if (V % 4 || (long long)B * T * (V / 4) >= (1LL << 31)) return false;
    for (const void* p : ptrs) if (p && !aligned16(p)) return false;
    return true;
}

int bn_steps_fwd_f32(const float* x, const float* gamma, const float* beta, float eps, float momentum, int B, int T, int V, float* y,
                     float* mean, float* rstd, float* s, float* b, float* uvar, float* run_mean, float* run_var, hipStream_t st) {
    if (!x || !y || !mean || !rstd || !s || !b || !uvar || B <= 0 || T <= 0 || V <= 0) { set_error("bn_steps_fwd: bad arguments"); return ICAP_ERR_ARG; }
    if (!bn_steps_ok(B, T, V, {x, y, gamma, beta, mean, rstd, s, b, uvar})) {
        set_error("bn_steps_fwd: V %% 4 == 0, fewer than 2^31 channel quads and 16-byte aligned operands required");
        return ICAP_ERR_ALIGN;
    }
    const int V4 = V / 4;
    hipLaunchKernelGGL(bn_steps_stats_kernel, dim3((V4 + BNS_TX - 1) / BNS_TX, T), dim3(256), 0, st, x, gamma, beta, eps, B, V, mean, rstd, s, b, uvar);
    if (run_mean || run_var)
        hipLaunchKernelGGL(bn_steps_running_kernel, dim3((V + 255) / 256), dim3(256), 0, st, mean, uvar, momentum, T, V, run_mean, run_var);
    long long blocks = ((long long)T * B * V4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bn_steps_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, s, b, B, T, V4, y);
    return launch_status("bn_steps_fwd");
}

int bn_steps_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd, const float* s, int B, int T, int V,
                     float* dgamma_t, float* dbeta_t, float* dx, hipStream_t st) {
    if (!dy || !x || !mean || !rstd || !s || !dgamma_t || !dbeta_t || !dx || B <= 0 || T <= 0 || V <= 0) { set_error("bn_steps_bwd: bad arguments"); return ICAP_ERR_ARG; }
    if (!bn_steps_ok(B, T, V, {dy, x, mean, rstd, s, dgamma_t, dbeta_t, dx})) {
        set_error("bn_steps_bwd: V %% 4 == 0, fewer than 2^31 channel quads and 16-byte aligned operands required");
        return ICAP_ERR_ALIGN;
    }
    const int V4 = V / 4;
    hipLaunchKernelGGL(bn_steps_bwd_sums_kernel, dim3((V4 + BNS_TX - 1) / BNS_TX, T), dim3(256), 0, st, dy, x, mean, rstd, B, T, V, dgamma_t, dbeta_t);
    long long blocks = ((long long)T * B * V4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(bn_steps_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, x, mean, rstd, s, dgamma_t, dbeta_t, B, T, V4, dx);
    return launch_status("bn_steps_bwd");
}

// Gradient of y = act(BatchNorm_train(x)) when the incoming gradient still has to pass the squeeze-excite gate:
//   dy_eff[b,p,c] = dy[b,p,c] * gate[b][c] + dpool[b][c] * alpha        (rows = B images of HW positions, dense [M][C])
// Both passes (the two column sums, then dx) read dy and apply the gate on the fly; dy_eff is never written.
// dgamma / dbeta [C] are results.  ws: col_reduce_ws_floats(M, C) floats.
int bn_act_bwd_se_f32(const float* dy, const float* x, const float* s, const float* b, const float* mean, const float* rstd,
                      const float* gate, const float* dpool, float alpha, int HW, int act, int M, int C, float* dgamma, float* dbeta,
                      float* dx, float* ws, hipStream_t st) {
    if (!dy || !x || !s || !b || !mean || !rstd || !gate || !dpool || !dgamma || !dbeta || !dx || !ws || M <= 0 || C <= 0 || HW <= 0 || M % HW) {
        set_error("bn_act_bwd_se: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (!vec4_ok(M, C, {C}, {dy, x, dx, s, b, mean, rstd, gate, dpool, dgamma, dbeta, ws})) {
        set_error("bn_act_bwd_se: C %% 4 == 0, fewer than 2^31 channel quads and 16-byte aligned operands required");
        return ICAP_ERR_ALIGN;
    }
    const SeGate sg{gate, dpool, alpha, HW};
    const int c4 = C / 4, tx = col_vec_tx(c4), rpc = col_rows_per_chunk(M, C), nchunks = (M + rpc - 1) / rpc;
    const dim3 grid((c4 + tx - 1) / tx, nchunks);
    if (act == ACT_SWISH) col_launch_vec<COL_BN_BWD, ACT_SWISH, true>(tx, grid, st, dy, C, x, C, mean, rstd, s, b, nullptr, nullptr, act, M, C, rpc, ws, sg);
    else col_launch_vec<COL_BN_BWD, -1, true>(tx, grid, st, dy, C, x, C, mean, rstd, s, b, nullptr, nullptr, act, M, C, rpc, ws, sg);
    hipLaunchKernelGGL(col_finish_kernel, dim3((C + 7) / 8), dim3(256), 0, st, ws, nchunks, C, 2, 1.0f, 0, dbeta, dgamma);
    int gcd = c4, t256 = 256;
    while (t256) { const int t = gcd % t256; gcd = t256; t256 = t; }
    const long long mult = c4 / gcd;
    long long vb = ((long long)M * c4 + 256LL * BWD_U * 4 - 1) / (256LL * BWD_U * 4);
    if (vb > 16384) vb = 16384;
    vb = (vb + mult - 1) / mult * mult;
    if (act == ACT_SWISH)
        hipLaunchKernelGGL((bn_act_bwd_vec_kernel<ACT_SWISH, true>), dim3((unsigned)vb), dim3(256), 0, st, dy, C, x, C, s, b, mean, rstd, dgamma, dbeta,
                           act, M, c4, dx, C, sg);
    else
        hipLaunchKernelGGL((bn_act_bwd_vec_kernel<-1, true>), dim3((unsigned)vb), dim3(256), 0, st, dy, C, x, C, s, b, mean, rstd, dgamma, dbeta,
                           act, M, c4, dx, C, sg);
    return launch_status("bn_act_bwd_se");
}

// ---------------------------------------------------------------- elementwise helpers
// y = alpha * x + beta * y over n elements
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float alpha, float beta, float* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = alpha * x[i] + (beta != 0.f ? beta * y[i] : 0.f);
}

int axpby_f32(const float* x, float alpha, float beta, float* y, long long n, hipStream_t st) {
    if (!x || !y || n <= 0) { set_error("axpby: bad arguments"); return ICAP_ERR_ARG; }
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, alpha, beta, y, n);
    return launch_status("axpby");
}

// Dropout with a counter-based generator: element i of call (seed, offset) keeps its value, scaled by 1/(1-p),
// when hash(seed, offset + i) >= p.  mask[i] holds the factor that was applied (0 or 1/(1-p)), which is also the
// gradient factor.  x may equal y.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float p, unsigned long long seed,
                                                      unsigned long long offset, long long n, long long group,
                                                      float* __restrict__ y, float* __restrict__ mask) {
    const float keep = 1.0f / (1.0f - p);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        // group > 1: one decision per group of consecutive elements (drop-connect: per sample)
        const float f = uniform01(seed, offset + (unsigned long long)(i / group)) >= p ? keep : 0.f;
        mask[group > 1 ? i / group : i] = f;
        y[i] = x[i] * f;
    }
}

// The same with the offset split in two: `offset` (a launch argument) + *offset_base (device memory).  A launch recorded
// in a HIP graph keeps its arguments for ever; the base is what moves from one replay to the next.
__global__ __launch_bounds__(256) void dropout_dev_kernel(const float* __restrict__ x, float p, unsigned long long seed,
                                                          unsigned long long offset, const unsigned long long* __restrict__ offset_base,
                                                          long long n, long long group, float* __restrict__ y, float* __restrict__ mask) {
    const float keep = 1.0f / (1.0f - p);
    const unsigned long long off = offset + *offset_base;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float f = uniform01(seed, off + (unsigned long long)(i / group)) >= p ? keep : 0.f;
        mask[group > 1 ? i / group : i] = f;
        y[i] = x[i] * f;
    }
}

int dropout_dev_f32(const float* x, float p, unsigned long long seed, unsigned long long offset, const unsigned long long* offset_base,
                    long long n, long long group, float* y, float* mask, hipStream_t st) {
    if (!x || !y || !mask || !offset_base || n <= 0 || group <= 0 || !(p >= 0.f && p < 1.f)) { set_error("dropout_dev: bad arguments"); return ICAP_ERR_ARG; }
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dropout_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, p, seed, offset, offset_base, n, group, y, mask);
    return launch_status("dropout_dev");
}

int dropout_f32(const float* x, float p, unsigned long long seed, unsigned long long offset, long long n, long long group,
                float* y, float* mask, hipStream_t st) {
    if (!x || !y || !mask || n <= 0 || group <= 0 || !(p >= 0.f && p < 1.f)) { set_error("dropout: bad arguments"); return ICAP_ERR_ARG; }
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, p, seed, offset, n, group, y, mask);
    return launch_status("dropout");
}

// y[i] = x[i] * mask[i / group] (+ add[i]): applies a stored dropout / drop-connect mask, forward or backward
__global__ __launch_bounds__(256) void mul_mask_kernel(const float* __restrict__ x, const float* __restrict__ mask, long long group,
                                                       const float* __restrict__ add, long long n, float* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = x[i] * mask[i / group];
        if (add) v += add[i];
        y[i] = v;
    }
}

int mul_mask_f32(const float* x, const float* mask, long long group, const float* add, long long n, float* y, hipStream_t st) {
    if (!x || !mask || !y || n <= 0 || group <= 0) { set_error("mul_mask: bad arguments"); return ICAP_ERR_ARG; }
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mul_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, mask, group, add, n, y);
    return launch_status("mul_mask");
}

// ---------------------------------------------------------------- LayerNorm (training): forward keeps mean / rstd
// y = ((x (+ res)) * pre_scale - mean) * rstd * gamma + beta, one wave per row; xs (optional) receives the
// normalised-input x (+ res) * pre_scale that the backward pass needs.
__global__ __launch_bounds__(256) void ln_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float pre_scale, int R, int E, float* __restrict__ y,
                                                           float* __restrict__ xs, float* __restrict__ mean_out,
                                                           float* __restrict__ rstd_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= R) return;
    const float* xr = x + (size_t)r * E;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) {
        float t = xr[e];
        if (res) t += res[(size_t)r * E + e];
        s += t * pre_scale;
    }
    const float mean = wave_sum(s) / (float)E;
    float q = 0.f;
    for (int e = lane; e < E; e += 64) {
        float t = xr[e];
        if (res) t += res[(size_t)r * E + e];
        const float d = t * pre_scale - mean;
        q = fmaf(d, d, q);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)E + eps);
    for (int e = lane; e < E; e += 64) {
        float t = xr[e];
        if (res) t += res[(size_t)r * E + e];
        t *= pre_scale;
        if (xs) xs[(size_t)r * E + e] = t;
        y[(size_t)r * E + e] = (t - mean) * rstd * gamma[e] + beta[e];
    }
    if (lane == 0) { mean_out[r] = mean; rstd_out[r] = rstd; }
}

int ln_train_fwd_f32(const float* x, const float* res, const float* gamma, const float* beta, float eps, float pre_scale, int R,
                     int E, float* y, float* xs, float* mean, float* rstd, hipStream_t st) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || R <= 0 || E <= 0) { set_error("ln_train_fwd: bad arguments"); return ICAP_ERR_ARG; }
    hipLaunchKernelGGL(ln_train_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, st, x, res, gamma, beta, eps, pre_scale, R, E, y, xs, mean, rstd);
    return launch_status("ln_train_fwd");
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) * post_scale,  g = dy * gamma,  xhat = (xs - mean) * rstd
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xs,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float post_scale, int R, int E,
                                                     float* __restrict__ dx) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= R) return;
    const float m = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
    for (int e = lane; e < E; e += 64) {
        const float g = dy[(size_t)r * E + e] * gamma[e];
        s1 += g;
        s2 += g * ((xs[(size_t)r * E + e] - m) * rs);
    }
    s1 = wave_sum(s1) / (float)E;
    s2 = wave_sum(s2) / (float)E;
    for (int e = lane; e < E; e += 64) {
        const float g = dy[(size_t)r * E + e] * gamma[e];
        const float xh = (xs[(size_t)r * E + e] - m) * rs;
        dx[(size_t)r * E + e] = rs * (g - s1 - xh * s2) * post_scale;
    }
}

int ln_bwd_f32(const float* dy, const float* xs, const float* gamma, const float* mean, const float* rstd, float post_scale, int R,
               int E, float* dx, hipStream_t st) {
    if (!dy || !xs || !gamma || !mean || !rstd || !dx || R <= 0 || E <= 0) { set_error("ln_bwd: bad arguments"); return ICAP_ERR_ARG; }
    hipLaunchKernelGGL(ln_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, st, dy, xs, gamma, mean, rstd, post_scale, R, E, dx);
    return launch_status("ln_bwd");
}

// ---------------------------------------------------------------- LSTM cell gradient
// gates: pre-activations [B][4H] (i, f, g, o); given dh, dc (gradient flowing into c_out from the next step):
//   do = dh * tanh(c_out);  dc_tot = dc + dh * o * (1 - tanh(c_out)^2)
//   di = dc_tot * g;  df = dc_tot * c_in;  dg = dc_tot * i;  dc_prev = dc_tot * f
// and the pre-activation gradients dgates = d* times the activation derivatives.
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_in,
                                                       const float* __restrict__ c_out, const float* __restrict__ dh,
                                                       const float* __restrict__ dc, int B, int H,
                                                       float* __restrict__ dgates, float* __restrict__ dc_prev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* g = gates + (size_t)b * 4 * H;
    const float ig = sigmoidf_(g[j]), fg = sigmoidf_(g[H + j]), gg = tanhf(g[2 * H + j]), og = sigmoidf_(g[3 * H + j]);
    const float tc = tanhf(c_out[i]);
    const float dhv = dh[i];
    const float dct = (dc ? dc[i] : 0.f) + dhv * og * (1.f - tc * tc);
    float* d = dgates + (size_t)b * 4 * H;
    d[j] = dct * gg * ig * (1.f - ig);
    d[H + j] = dct * c_in[i] * fg * (1.f - fg);
    d[2 * H + j] = dct * ig * (1.f - gg * gg);
    d[3 * H + j] = dhv * tc * og * (1.f - og);
    dc_prev[i] = dct * fg;
}

int lstm_bwd_f32(const float* gates, const float* c_in, const float* c_out, const float* dh, const float* dc, int B, int H,
                 float* dgates, float* dc_prev, hipStream_t st) {
    if (!gates || !c_in || !c_out || !dh || !dgates || !dc_prev || B <= 0 || H <= 0) { set_error("lstm_bwd: bad arguments"); return ICAP_ERR_ARG; }
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, gates, c_in, c_out, dh, dc, B, H, dgates, dc_prev);
    return launch_status("lstm_bwd");
}

// ---------------------------------------------------------------- GRU cell gradient
// Forward (gate order r, z, n; gi = W_ih x + b_ih, gh = W_hh h + b_hh, both [B][3H]):
//   r = sig(gi_r + gh_r), z = sig(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h.
// Given dh': dgi, dgh (pre-activation gradients of the two halves) and the direct path dh_direct = dh' * z.
__global__ __launch_bounds__(256) void gru_bwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                      const float* __restrict__ h_in, const float* __restrict__ dh, int B, int H,
                                                      float* __restrict__ dgi, float* __restrict__ dgh,
                                                      float* __restrict__ dh_direct) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, j = i - b * H;
    const float* a = gi + (size_t)b * 3 * H;
    const float* c = gh + (size_t)b * 3 * H;
    const float r = sigmoidf_(a[j] + c[j]);
    const float z = sigmoidf_(a[H + j] + c[H + j]);
    const float ghn = c[2 * H + j];
    const float n = tanhf(a[2 * H + j] + r * ghn);
    const float d = dh[i];
    const float dpn = d * (1.f - z) * (1.f - n * n);
    const float dpr = dpn * ghn * r * (1.f - r);
    const float dpz = d * (h_in[i] - n) * z * (1.f - z);
    float* oa = dgi + (size_t)b * 3 * H;
    float* oc = dgh + (size_t)b * 3 * H;
    oa[j] = dpr; oc[j] = dpr;
    oa[H + j] = dpz; oc[H + j] = dpz;
    oa[2 * H + j] = dpn; oc[2 * H + j] = dpn * r;
    dh_direct[i] = d * z;
}

int gru_bwd_f32(const float* gi, const float* gh, const float* h_in, const float* dh, int B, int H, float* dgi, float* dgh,
                float* dh_direct, hipStream_t st) {
    if (!gi || !gh || !h_in || !dh || !dgi || !dgh || !dh_direct || B <= 0 || H <= 0) { set_error("gru_bwd: bad arguments"); return ICAP_ERR_ARG; }
    hipLaunchKernelGGL(gru_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, st, gi, gh, h_in, dh, B, H, dgi, dgh, dh_direct);
    return launch_status("gru_bwd");
}

// ---------------------------------------------------------------- additive attention gradient
// Forward (transformer_word_level_bahdnau.py:102-110): score[i][j] = a_w . tanh(u[j] + w[i]) + a_b,
// alpha[i] = softmax_j(score[i]), context[i] = sum_j alpha[i][j] feat[j].
// Query side, one workgroup per i:  dalpha[i][j] = dctx[i] . feat[j];
//   dscore[i][j] = alpha[i][j] * (dalpha[i][j] - sum_j' alpha[i][j'] dalpha[i][j']);
//   dw[i][a] = sum_j dscore[i][j] * a_w[a] * (1 - tanh(u[j][a] + w[i][a])^2);
//   da_part[i][a] = sum_j dscore[i][j] * tanh(u[j][a] + w[i][a]);   dab_part[i] = sum_j dscore[i][j].
// Both kernels run 16 waves per workgroup: the long sums over the other batch index are split four ways across the
// workgroup (thread (q, t) takes a quarter of the range for column t) and combined through LDS in a fixed order, and
// the inner loops are unrolled so that the loads of several iterations are in flight before the first tanh.
constexpr int ATT_MAX_B = 1024;
constexpr int ATT_THREADS = 1024;

__global__ __launch_bounds__(ATT_THREADS) void att_bwd_query_kernel(const float* __restrict__ u, const float* __restrict__ w,
                                                                    const float* __restrict__ a_w, const float* __restrict__ alpha,
                                                                    const float* __restrict__ feat, const float* __restrict__ dctx,
                                                                    int B, int A, int E, float* __restrict__ dscore,
                                                                    float* __restrict__ dw, float* __restrict__ da_part,
                                                                    float* __restrict__ dab_part) {
    __shared__ float ds[ATT_MAX_B];
    __shared__ float red[16];
    __shared__ float pw[4][256], pa[4][256];
    const int i = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int j0 = wave; j0 < B; j0 += 32) {      // two keys per wave and pass
        const int j1 = j0 + 16 < B ? j0 + 16 : j0;
        float s0 = 0.f, s1 = 0.f;
        for (int e = lane; e < E; e += 64) {
            const float d = dctx[(size_t)i * E + e];
            s0 = fmaf(d, feat[(size_t)j0 * E + e], s0);
            s1 = fmaf(d, feat[(size_t)j1 * E + e], s1);
        }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) {
            ds[j0] = s0;   // dalpha
            if (j0 + 16 < B) ds[j0 + 16] = s1;
        }
    }
    __syncthreads();
    float dot = 0.f;
    for (int j = tid; j < B; j += ATT_THREADS) dot += alpha[(size_t)i * B + j] * ds[j];
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) dot += red[k];
    __syncthreads();
    float tot = 0.f;
    for (int j = tid; j < B; j += ATT_THREADS) {
        const float v = alpha[(size_t)i * B + j] * (ds[j] - dot);
        ds[j] = v;
        dscore[(size_t)i * B + j] = v;
        tot += v;
    }
    tot = wave_sum(tot);
    if (lane == 0) red[wave] = tot;
    __syncthreads();
    if (tid == 0) {
        float r = 0.f;
        for (int k = 0; k < 16; ++k) r += red[k];
        dab_part[i] = r;
    }
    const int q = tid >> 8, t = tid & 255;
    const int Bq = (B + 3) / 4, jlo = q * Bq, jhi = (jlo + Bq < B) ? jlo + Bq : B;
    for (int a0 = 0; a0 < A; a0 += 256) {
        const int a = a0 + t;
        float accw = 0.f, acca = 0.f;
        if (a < A) {
            const float wi = w[(size_t)i * A + a];
#pragma unroll 8
            for (int j = jlo; j < jhi; ++j) {
                const float th = tanh_fast(u[(size_t)j * A + a] + wi);
                accw += ds[j] * (1.f - th * th);
                acca += ds[j] * th;
            }
        }
        pw[q][t] = accw;
        pa[q][t] = acca;
        __syncthreads();
        if (q == 0 && a < A) {
            dw[(size_t)i * A + a] = (((pw[0][t] + pw[1][t]) + pw[2][t]) + pw[3][t]) * a_w[a];
            da_part[(size_t)i * A + a] = ((pa[0][t] + pa[1][t]) + pa[2][t]) + pa[3][t];
        }
        __syncthreads();
    }
}

// Key side, one workgroup per j:  du[j][a] (+)= sum_i dscore[i][j] * a_w[a] * (1 - tanh(u[j][a] + w[i][a])^2);
//   dfeat[j][e] (+)= sum_i alpha[i][j] * dctx[i][e].
__global__ __launch_bounds__(ATT_THREADS) void att_bwd_key_kernel(const float* __restrict__ u, const float* __restrict__ w,
                                                                  const float* __restrict__ a_w, const float* __restrict__ alpha,
                                                                  const float* __restrict__ dscore, const float* __restrict__ dctx,
                                                                  int B, int A, int E, int accumulate, float* __restrict__ du,
                                                                  float* __restrict__ dfeat) {
    __shared__ float dsj[ATT_MAX_B], alj[ATT_MAX_B];      // column j of dscore and alpha
    __shared__ float part[4][256];
    const int j = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < B; i += ATT_THREADS) {
        dsj[i] = dscore[(size_t)i * B + j];
        alj[i] = alpha[(size_t)i * B + j];
    }
    __syncthreads();
    const int q = tid >> 8, t = tid & 255;
    const int Bq = (B + 3) / 4, ilo = q * Bq, ihi = (ilo + Bq < B) ? ilo + Bq : B;
    for (int a0 = 0; a0 < A; a0 += 256) {
        const int a = a0 + t;
        float acc = 0.f;
        if (a < A) {
            const float uj = u[(size_t)j * A + a];
#pragma unroll 8
            for (int i = ilo; i < ihi; ++i) {
                const float th = tanh_fast(uj + w[(size_t)i * A + a]);
                acc += dsj[i] * (1.f - th * th);
            }
        }
        part[q][t] = acc;
        __syncthreads();
        if (q == 0 && a < A) {
            const size_t o = (size_t)j * A + a;
            du[o] = (accumulate ? du[o] : 0.f) + (((part[0][t] + part[1][t]) + part[2][t]) + part[3][t]) * a_w[a];
        }
        __syncthreads();
    }
    for (int e0 = 0; e0 < E; e0 += 256) {
        const int e = e0 + t;
        float acc = 0.f;
        if (e < E) {
#pragma unroll 8
            for (int i = ilo; i < ihi; ++i) acc += alj[i] * dctx[(size_t)i * E + e];
        }
        part[q][t] = acc;
        __syncthreads();
        if (q == 0 && e < E) {
            const size_t o = (size_t)j * E + e;
            dfeat[o] = (accumulate ? dfeat[o] : 0.f) + (((part[0][t] + part[1][t]) + part[2][t]) + part[3][t]);
        }
        __syncthreads();
    }
}

int att_bwd_f32(const float* u, const float* w, const float* a_w, const float* alpha, const float* feat, const float* dctx, int B,
                int A, int E, int accumulate, float* dscore, float* dw, float* da_part, float* dab_part, float* du, float* dfeat,
                hipStream_t st) {
    if (!u || !w || !a_w || !alpha || !feat || !dctx || !dscore || !dw || !da_part || !dab_part || !du || !dfeat || B <= 0 ||
        A <= 0 || E <= 0) {
        set_error("att_bwd: bad arguments");
        return ICAP_ERR_ARG;
    }
    if (B > ATT_MAX_B) { set_error("att_bwd: batch %d exceeds %d", B, ATT_MAX_B); return ICAP_ERR_LIMIT; }
    hipLaunchKernelGGL(att_bwd_query_kernel, dim3(B), dim3(ATT_THREADS), 0, st, u, w, a_w, alpha, feat, dctx, B, A, E, dscore, dw, da_part, dab_part);
    hipLaunchKernelGGL(att_bwd_key_kernel, dim3(B), dim3(ATT_THREADS), 0, st, u, w, a_w, alpha, dscore, dctx, B, A, E, accumulate, du, dfeat);
    return launch_status("att_bwd");
}

// ---------------------------------------------------------------- multi-head attention gradient
// Forward per head h (width D = E / heads): S = scale * Q_h K_h^T  [R x Bk], P = softmax_rows(S), O_h = P V_h.
// q, dO: [R][E]; k, v: [Bk][E].  Query side, one wave per (query r, head h): recompute P, then
//   dP_j = dO_r . v_j,  dS_j = P_j (dP_j - sum_j' P_j' dP_j'),  dQ_r = scale * sum_j dS_j k_j;
// P and dS ([R][heads][Bk]) are kept for the key side.
constexpr int MHA_MAX_KEYS = 1024;
constexpr int MHA_MAXD = 4;   // head width <= 256

__global__ __launch_bounds__(64) void mha_bwd_query_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ dO, int Bk,
                                                           int E, int heads, float scale, float* __restrict__ dq,
                                                           float* __restrict__ P, float* __restrict__ dS) {
    __shared__ float sc[MHA_MAX_KEYS];
    __shared__ float dp[MHA_MAX_KEYS];
    const int r = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const int D = E / heads;
    const size_t qo = (size_t)r * E + (size_t)h * D;
    float qr[MHA_MAXD], gr[MHA_MAXD], acc[MHA_MAXD];
#pragma unroll
    for (int i = 0; i < MHA_MAXD; ++i) {
        const int d = lane + 64 * i;
        qr[i] = d < D ? q[qo + d] : 0.f;
        gr[i] = d < D ? dO[qo + d] : 0.f;
        acc[i] = 0.f;
    }
    float m = -INFINITY;
    for (int j = 0; j < Bk; ++j) {
        const size_t ko = (size_t)j * E + (size_t)h * D;
        float s = 0.f, g = 0.f;
#pragma unroll
        for (int i = 0; i < MHA_MAXD; ++i) {
            const int d = lane + 64 * i;
            if (d < D) { s = fmaf(qr[i], k[ko + d], s); g = fmaf(gr[i], v[ko + d], g); }
        }
        s = wave_sum(s) * scale;
        g = wave_sum(g);
        if (lane == 0) { sc[j] = s; dp[j] = g; }
        m = fmaxf(m, s);
    }
    __syncthreads();
    float l = 0.f;
    for (int j = lane; j < Bk; j += 64) l += expf(sc[j] - m);
    l = wave_sum(l);
    const float inv = 1.0f / l;
    float delta = 0.f;
    for (int j = lane; j < Bk; j += 64) {
        const float pj = expf(sc[j] - m) * inv;
        sc[j] = pj;
        delta += pj * dp[j];
    }
    delta = wave_sum(delta);
    __syncthreads();
    const size_t po = ((size_t)r * heads + h) * Bk;
    for (int j = lane; j < Bk; j += 64) {
        const float ds = sc[j] * (dp[j] - delta);
        P[po + j] = sc[j];
        dS[po + j] = ds;
        dp[j] = ds;
    }
    __syncthreads();
    for (int j = 0; j < Bk; ++j) {
        const size_t ko = (size_t)j * E + (size_t)h * D;
        const float ds = dp[j];
#pragma unroll
        for (int i = 0; i < MHA_MAXD; ++i) {
            const int d = lane + 64 * i;
            if (d < D) acc[i] = fmaf(ds, k[ko + d], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < MHA_MAXD; ++i) {
        const int d = lane + 64 * i;
        if (d < D) dq[qo + d] = acc[i] * scale;
    }
}

// Key side, one workgroup per (key j, head h), thread = channel of the head:
//   dK_j = scale * sum_r dS[r][h][j] q_r,   dV_j = sum_r P[r][h][j] dO_r   (r in order: deterministic).
__global__ __launch_bounds__(256) void mha_bwd_key_kernel(const float* __restrict__ q, const float* __restrict__ dO,
                                                          const float* __restrict__ P, const float* __restrict__ dS, int R, int Bk,
                                                          int E, int heads, float scale, float* __restrict__ dk,
                                                          float* __restrict__ dv) {
    const int j = blockIdx.x, h = blockIdx.y, d = threadIdx.x;
    const int D = E / heads;
    if (d >= D) return;
    float ak = 0.f, av = 0.f;
    for (int r = 0; r < R; ++r) {
        const size_t po = ((size_t)r * heads + h) * Bk + j;
        const size_t qo = (size_t)r * E + (size_t)h * D + d;
        ak = fmaf(dS[po], q[qo], ak);
        av = fmaf(P[po], dO[qo], av);
    }
    const size_t ko = (size_t)j * E + (size_t)h * D + d;
    dk[ko] = ak * scale;
    dv[ko] = av;
}

int mha_bwd_f32(const float* q, const float* k, const float* v, const float* dO, int R, int Bk, int E, int heads, float* dq,
                float* dk, float* dv, float* ws, hipStream_t st) {
    if (!q || !k || !v || !dO || !dq || !dk || !dv || !ws || R <= 0 || Bk <= 0 || E <= 0 || heads <= 0 || E % heads) {
        set_error("mha_bwd: bad arguments");
        return ICAP_ERR_ARG;
    }
    const int D = E / heads;
    if (Bk > MHA_MAX_KEYS || D > 64 * MHA_MAXD) { set_error("mha_bwd: %d keys / head width %d exceed %d / %d", Bk, D, MHA_MAX_KEYS, 64 * MHA_MAXD); return ICAP_ERR_LIMIT; }
    const float scale = 1.0f / sqrtf((float)D);
    float* P = ws;
    float* dS = ws + (size_t)R * heads * Bk;
    hipLaunchKernelGGL(mha_bwd_query_kernel, dim3(R, heads), dim3(64), 0, st, q, k, v, dO, Bk, E, heads, scale, dq, P, dS);
    hipLaunchKernelGGL(mha_bwd_key_kernel, dim3(Bk, heads), dim3(256), 0, st, q, dO, P, dS, R, Bk, E, heads, scale, dk, dv);
    return launch_status("mha_bwd");
}

// ---------------------------------------------------------------- embedding gradient
// dtable[idx[p]][:] += dy[p][:] for p in order.  Workgroup g owns the table rows with row % EMB_GROUPS == g, so
// every row is updated by one workgroup in position order: deterministic without atomics.  A workgroup looks at 1024
// positions at a time, compacts the ones it owns into an LDS list (wave ballots keep them in position order) and then
// applies them in that order, each thread adding its own columns (runs of one table row summed in a register first).  dtable must be zeroed (or hold the value to
// accumulate onto) before the call.
constexpr int EMB_GROUPS = 256;      // power of two
constexpr int EMB_THREADS = 1024;

__global__ __launch_bounds__(EMB_THREADS) void embedding_bwd_kernel(const long long* __restrict__ idx, const float* __restrict__ dy,
                                                                    int ldd, int P, int E, int V, float* __restrict__ dtable) {
    __shared__ int list_p[EMB_THREADS];
    __shared__ int list_t[EMB_THREADS];
    __shared__ int wc[EMB_THREADS / 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int c0 = 0; c0 < P; c0 += EMB_THREADS) {
        const int p = c0 + tid;
        long long t = -1;
        if (p < P) t = idx[p];
        const bool mine = t >= 0 && t < V && (int)(t & (EMB_GROUPS - 1)) == (int)blockIdx.x;
        const unsigned long long bal = __ballot(mine);
        if (lane == 0) wc[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int k = 0; k < EMB_THREADS / 64; ++k) {
            const int c = wc[k];
            off += k < wave ? c : 0;
            total += c;
        }
        if (mine) {
            const int o = off + __popcll(bal & ((1ull << lane) - 1ull));
            list_p[o] = p;
            list_t[o] = (int)t;
        }
        __syncthreads();
        // a thread owns columns tid, tid + 1024, ...; consecutive list entries for the same table row (the padding token at
        // the late time steps comes in long runs) are summed in a register and written once - entry by entry the loop is a
        // chain of read-modify-write round trips through memory.  Four gradient rows are loaded ahead of their use.
        for (int e = tid; e < E; e += EMB_THREADS) {
            int cur = -1;
            float acc = 0.f;
            for (int k0 = 0; k0 < total; k0 += 4) {
                float v[4];
                int tr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = min(k0 + u, total - 1);
                    tr[u] = list_t[k];
                    v[u] = dy[(size_t)list_p[k] * ldd + e];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k0 + u >= total) break;
                    if (tr[u] != cur) {
                        if (cur >= 0) dtable[(size_t)cur * E + e] += acc;
                        cur = tr[u];
                        acc = 0.f;
                    }
                    acc += v[u];
                }
            }
            if (cur >= 0) dtable[(size_t)cur * E + e] += acc;
        }
        __syncthreads();
    }
}

int embedding_bwd_f32(const long long* idx, const float* dy, int ldd, int P, int E, int V, float* dtable, hipStream_t st) {
    if (!idx || !dy || !dtable || P <= 0 || E <= 0 || V <= 0 || ldd < E) { set_error("embedding_bwd: bad arguments"); return ICAP_ERR_ARG; }
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3(EMB_GROUPS), dim3(EMB_THREADS), 0, st, idx, dy, ldd, P, E, V, dtable);
    return launch_status("embedding_bwd");
}

}  // namespace icap
