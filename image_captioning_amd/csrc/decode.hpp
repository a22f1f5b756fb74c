// Host-side interface of the small-batch decode kernels (decode.hip), shared with
// the greedy drivers in capi.hip.
#pragma once
#include "common.hpp"

namespace icap {

// How a decode kernel obtains its [NB][K] input vectors.  Every workgroup builds the
// vectors in LDS itself, so the token selection, the embedding-row gather and the
// LayerNorm that precede a weight-streaming kernel cost no launch of their own.
//
//   plain:  v[b][k] = x[b*ldx + k]                                   (table == null)
//   token:  tok[b]  = argmax over part_val/part_idx[b][0..nparts)     (part_val != null)
//                     or tok_in[b]                                    (otherwise)
//           v[b][k] = table[tok[b]*E + k]            k <  E
//                   = tail[b*ld_tail + k - E]        k >= E
//   then    v[b][k] += add[b*ld_add + k]            (add != null; k < E in token mode)
//   then    v[b][:]  = LayerNorm(v[b][:]) * ln_g + ln_b               (ln_g != null)
// Workgroup 0 stores tok[b] to ids_out[b*ids_stride] and v to x_out when those are set.
struct StepInput {
    const float* x = nullptr;
    int ldx = 0;
    const float* table = nullptr;
    int E = 0;
    int table_rows = 0;
    const float* part_val = nullptr;
    const int* part_idx = nullptr;
    int nparts = 0;
    const long long* tok_in = nullptr;
    long long* ids_out = nullptr;
    int ids_stride = 0;
    const float* tail = nullptr;
    int ld_tail = 0;
    const float* add = nullptr;
    int ld_add = 0;
    const float* ln_g = nullptr;
    const float* ln_b = nullptr;
    float eps = 1e-5f;
    float* x_out = nullptr;
    int ld_xout = 0;
};

inline StepInput plain_input(const float* x, int ldx) {
    StepInput s;
    s.x = x;
    s.ldx = ldx;
    return s;
}

// w16: the weight matrices are bf16 (icap_to_bf16_f32 of the fp32 ones, same row-major layout, K / I / H multiples of 8);
// everything else - inputs, biases, accumulation, outputs - stays fp32.
int gemv_step_f32(const void* W, const float* scale, const float* bias, const StepInput& in, float* y, int ldy, int NB,
                  int N, int K, int act, float* part_val, int* part_idx, hipStream_t st, bool w16 = false);
int rnn_cell_step_f32(bool lstm, const void* w_ih, const void* w_hh, const float* b_ih, const float* b_hh,
                      const StepInput& in, const float* h_in, const float* c_in, float* h_out, float* c_out, int NB, int I,
                      int H, hipStream_t st, bool w16 = false);
int gemv_blocks(int N);
int argmax_finalize_f32(const float* part_val, const int* part_idx, int nparts, long long* ids, int ids_stride,
                        const float* table, const float* add, float* next_x, int ldx, int E, int NB, hipStream_t st);

}  // namespace icap
