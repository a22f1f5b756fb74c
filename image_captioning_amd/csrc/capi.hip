// extern "C" surface of libicap_hip.so (declared in include/icap_hip.h) and the
// greedy decode drivers, which enqueue a whole caption's worth of step kernels on
// the caller's stream without touching the host in between.
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include "decode.hpp"
#include "../../include/icap_hip.h"

namespace icap {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// kernels implemented in the other translation units
int conv_igemm_f32(const float*, const float*, const float*, const float*, const float*, const float*, float*, int, int,
                   int, int, int, int, int, int, int, int, int, int, int, hipStream_t);
int conv_igemm_tune_f32(const float*, const float*, const float*, const float*, const float*, const float*, float*, int,
                        int, int, int, int, int, int, int, int, int, int, int, int, hipStream_t);
int conv_igemm_tuned_cfg(int, int, int, int, int, int, int, int, int, int);
int conv_igemm_num_cfgs();
int table_walk(const long long*, const long long*, long long*, const float*, int, float*, int, int, int, hipStream_t);
long long greedy_bah_fused_ws_floats(int, int, int);
int greedy_bah_fused_f32(const float*, const float*, const float*, const float*, const float*, const float*, const long long*, long long*,
                         float*, float*, int, int, int, int, hipStream_t);
long long greedy_rnn2_fused_ws_floats(int, int, int, int);
int greedy_rnn2_fused_f32(int, const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                          const float*, const float*, const float*, long long*, float*, float*, int, int, int, int, hipStream_t);
int greedy_tfm_persist_max_nb(int, int, int);
long long greedy_tfm_persist_ws_bytes(int, int, int, int);
int greedy_tfm_persist_f32(const icap_tfm_weights*, const float*, const long long*, long long*, float*, void*, int, int, hipStream_t);
int linear_thin_supported(int, int);
int linear_f32x3_tall_supported(int, int, int, int);
long long linear_f32x3_splitk_ws_floats(int, int, int);
int linear_f32x3_splitk_f32(const float*, const unsigned short*, const float*, const float*, const float*, const float*, int, float*, float*,
                            long long, int, int, int, int, int, int, hipStream_t);
int linear_thin_f32(const float*, const float*, const float*, const float*, const float*, const float*, int, float*, long long, int, int,
                    hipStream_t);
int greedy_bah_persist_max_nb(int, int);
int greedy_rnn2_persist_max_nb(int, int, int);
long long greedy_rnn2_persist_ws_bytes(int, int, int, int);
int greedy_rnn2_persist_f32(int, const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                            const float*, const float*, long long*, float*, void*, int, int, int, int, hipStream_t);
long long greedy_bah_persist_ws_bytes(int, int, int);
int greedy_bah_persist_f32(const float*, const float*, const float*, const float*, const float*, const float*, const long long*, long long*,
                           float*, void*, int, int, int, int, hipStream_t);
int conv_igemm_use_tuned(int);
int conv_igemm_cfg_f32(const float*, const float*, const float*, const float*, const float*, const float*, float*, int, int,
                       int, int, int, int, int, int, int, int, int, int, int, int, hipStream_t);
int linear_argmax_parts(int);
int linear_argmax_f32(const float*, const float*, const float*, const float*, float*, float*, int*, int, int, int, hipStream_t);
int next_token_f32(const float*, const int*, int, long long*, int, const float*, const float*, const float*, const float*, float,
                   float*, int, int, int, hipStream_t);
int linear_bf16_f32(const void*, int, const unsigned short*, const float*, const float*, const float*, const float*, int, float*, int,
                    int, int, int, int, hipStream_t);
int to_bf16_f32(const float*, unsigned short*, long long, hipStream_t);
int split3_f32(const float*, unsigned short*, long long, hipStream_t);
int linear_f32x3_f32(const float*, int, const unsigned short*, const float*, const float*, const float*, const float*, int, float*, int,
                     int, int, int, int, hipStream_t);
int linear_f32x3_argmax_f32(const float*, int, const unsigned short*, const float*, const float*, float*, float*, int*, int, int, int,
                            hipStream_t);
int linear_x3dl_f32(const unsigned short*, const unsigned short*, const float*, const float*, const float*, float*, unsigned short*, int, int,
                    int, int, int, hipStream_t);
int linear_x3dl_argmax_f32(const unsigned short*, const unsigned short*, const float*, const float*, float*, float*, int*, int, int, int,
                           hipStream_t);
int conv2d_bf16_f32(const void*, const unsigned short*, const float*, const float*, const void*, void*, int, int, int, int, int, int,
                    int, int, int, int, int, int, int, int, int, hipStream_t);
int maxpool2d_bf16(const unsigned short*, unsigned short*, int, int, int, int, int, int, int, int, int, hipStream_t);
int adaptive_avgpool2d_bf16_f32(const unsigned short*, float*, int, int, int, int, int, int, hipStream_t);
int linear_bf16_argmax_parts(int);
int linear_bf16_argmax_f32(const float*, const unsigned short*, const float*, const float*, float*, float*, int*, int, int, int, hipStream_t);
int topk_sample_f32(const float*, int, int, int, int, unsigned long long, unsigned long long, float*, int*, long long*, int, hipStream_t);
int nchw_to_nhwc_pad_f32(const float*, float*, int, int, int, int, int, hipStream_t);
long long resize_normalize_ws_bytes(int, int, int);
long long augment_ws_bytes(int);
int augment_u8(const unsigned char*, int, int, int, const int*, const float*, unsigned long long*, unsigned char*, hipStream_t);
int resize_normalize_u8_f32(const unsigned char*, int, int, int, const int*, const int*, int, const int*, const int*, int, int, int,
                            const float*, unsigned*, float*, int, hipStream_t);
int dwconv2d_f32(const float*, const float*, const float*, const float*, float*, float*, int, int, int, int, int, int,
                 int, int, int, int, int, hipStream_t);
int dwconv_chunks(int, int, int, int, int);
int expand_dwconv_f32(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                      float*, int, int, int, int, int, int, int, int, int, int, int, int, hipStream_t);
int expand_dwconv_chunks(int, int, int, int, int, int);
int expand_dwconv_whole_image(int, int, int, int, int, int);
long long expand_dwconv_img_x3_we_bytes(int, int, int, int, int);
int expand_dwconv_img_x3_pack(const float*, void*, int, int, int, int, int, hipStream_t);
int expand_dwconv_img_x3_f32(const float*, const void*, const float*, const float*, const float*, const float*, const float*, float*,
                             float*, int, int, int, int, int, int, int, int, hipStream_t);
int expand_dwconv_img_x3_s2_supported(int, int, int, int, int, int, int);
int expand_dwconv_img_x3_s2_f32(const float*, const void*, const float*, const float*, const float*, const float*, const float*, float*,
                                float*, int, int, int, int, int, int, int, int, hipStream_t);
int stem_dwconv_f32(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*,
                    float*, int, int, int, int, int, int, int, int, int, int, int, int, int, hipStream_t);
int stem_dwconv_chunks(int, int, int);
int stem_dwconv_valu_supported(int);
int stem_dwconv_valu_f32(const float*, const float*, const float*, const float*, const float*, const float*, const float*, float*, float*,
                         int, int, int, int, int, int, int, int, int, int, int, int, hipStream_t);
int se_gate_f32(const float*, int, float, const float*, const float*, const float*, const float*, float*, int, int, int,
                hipStream_t);
int maxpool2d_f32(const float*, float*, int, int, int, int, int, int, int, int, int, hipStream_t);
int adaptive_avgpool2d_f32(const float*, float*, int, int, int, int, int, int, hipStream_t);
int embedding_f32(const float*, const long long*, const float*, float*, int, int, int, int, int*, hipStream_t);
int layernorm_f32(const float*, const float*, int, const float*, const float*, float*, int, int, float, float, hipStream_t);
int layernorm_split3_f32(const float*, const float*, int, const float*, const float*, float*, unsigned short*, int, int, float, float,
                         hipStream_t);
int argmax_rows_f32(const float*, int, long long*, int, int, int, float*, hipStream_t);
int mha_core_f32(const float*, const float*, const float*, float*, int, int, int, int, int, int, hipStream_t);
int bahdanau_attention_f32(const float*, const float*, const float*, float, const float*, float*, float*, int, int, int,
                           hipStream_t);
int lstm_pointwise_f32(const float*, const float*, float*, float*, int, int, hipStream_t);
int gru_pointwise_f32(const float*, const float*, const float*, float*, int, int, hipStream_t);
int gemv_f32(const float*, const float*, const float*, const float*, int, float*, int, int, int, int, int, float*, int*,
             hipStream_t);
int lstm_cell_f32(const float*, const float*, const float*, const float*, const float*, int, const float*, const float*,
                  float*, float*, int, int, int, hipStream_t);
int gru_cell_f32(const float*, const float*, const float*, const float*, const float*, int, const float*, float*, int, int,
                 int, hipStream_t);

int ce_loss_fwd_f32(const float*, int, int, int, const long long*, int, int, long long, int, float*, float*, float*, hipStream_t);
int ce_loss_bwd_f32(const float*, int, int, int, const long long*, int, int, long long, const float*, const float*, const float*,
                    float*, int, hipStream_t);
int adamw_f32(float*, const float*, float*, float*, long long, float, float, float, float, float, int, float, hipStream_t);
int adamw_dev_f32(float*, const float*, float*, float*, long long, float, float, float, float, float, const int*, int, float, hipStream_t);

int gemm_bwd_f32(const float*, int, int, const float*, int, float*, int, int, int, int, int, float*, hipStream_t);
long long gemm_bwd_ws_floats(int, int, int, int);
int col_reduce_f32(int, const float*, int, const float*, int, const float*, const float*, const float*, const float*, const float*,
                   const float*, int, int, int, float, int, float*, float*, float*, hipStream_t);
long long col_reduce_ws_floats(int, int);
int bn_train_stats_f32(const float*, int, const float*, const float*, float, float, int, int, float*, float*, float*, float*, float*,
                       float*, float*, hipStream_t);
long long conv_stats_max_parts(long long);
int conv_igemm_stats_f32(const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int, int, int, float*, long long,
                         int*, hipStream_t);
int linear_f32x3_stats_f32(const float*, const unsigned short*, float*, int, int, int, int, float*, long long, int*, hipStream_t);
int dwconv2d_stats_f32(const float*, const float*, float*, int, int, int, int, int, int, int, int, int, int, float*, float*, hipStream_t);
long long bn_stats_from_parts_ws_floats(int, int);
int bn_stats_from_parts_f32(const float*, const float*, int, int, int, int, const float*, const float*, float, float, float*, float*,
                            float*, float*, float*, float*, float*, hipStream_t);
int bn_finalize_moments_f32(const float*, const float*, const float*, const float*, const float*, float, float, int, int, float*,
                            float*, float*, float*, float*, float*, hipStream_t);
int bn_finalize_f32(const float*, const float*, const float*, const float*, float, float, int, int, float*, float*, float*, float*,
                    float*, hipStream_t);
int scale_bias_act_f32(const float*, int, const float*, const float*, const float*, int, int, int, int, float*, int, hipStream_t);
int bn_act_bwd_f32(const float*, int, const float*, int, const float*, const float*, const float*, const float*, const float*,
                   const float*, int, int, int, float*, int, hipStream_t);
int axpby_f32(const float*, float, float, float*, long long, hipStream_t);
int dropout_f32(const float*, float, unsigned long long, unsigned long long, long long, long long, float*, float*, hipStream_t);
int dropout_dev_f32(const float*, float, unsigned long long, unsigned long long, const unsigned long long*, long long, long long, float*, float*,
                    hipStream_t);
int mul_mask_f32(const float*, const float*, long long, const float*, long long, float*, hipStream_t);
int ln_train_fwd_f32(const float*, const float*, const float*, const float*, float, float, int, int, float*, float*, float*, float*,
                     hipStream_t);
int ln_bwd_f32(const float*, const float*, const float*, const float*, const float*, float, int, int, float*, hipStream_t);
int lstm_bwd_f32(const float*, const float*, const float*, const float*, const float*, int, int, float*, float*, hipStream_t);
int gru_bwd_f32(const float*, const float*, const float*, const float*, int, int, float*, float*, float*, hipStream_t);
int att_bwd_f32(const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, float*,
                float*, float*, float*, float*, float*, hipStream_t);
int mha_bwd_f32(const float*, const float*, const float*, const float*, int, int, int, int, float*, float*, float*, float*, hipStream_t);
int embedding_bwd_f32(const long long*, const float*, int, int, int, int, float*, hipStream_t);

int chan_affine_f32(const float*, const float*, const float*, float, int, int, int, float*, hipStream_t);
int image_dot_f32(const float*, const float*, int, int, int, float, float*, float*, hipStream_t);
long long image_dot_ws_floats(int, int, int);
int act_pool_f32(const float*, const float*, const float*, int, int, int, int, float, float*, float*, float*, hipStream_t);
int bn_steps_fwd_f32(const float*, const float*, const float*, float, float, int, int, int, float*, float*, float*, float*, float*,
                     float*, float*, float*, hipStream_t);
int bn_steps_bwd_f32(const float*, const float*, const float*, const float*, const float*, int, int, int, float*, float*, float*,
                     hipStream_t);
long long act_pool_ws_floats(int, int, int);
int act_gate_f32(const float*, const float*, const float*, int, const float*, int, int, int, float*, hipStream_t);
int image_dot_act_f32(const float*, const float*, const float*, const float*, int, int, int, int, float, float*, float*, hipStream_t);
int bn_act_bwd_se_f32(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                      float, int, int, int, int, float*, float*, float*, float*, hipStream_t);

int se_train_fwd_f32(const float*, const float*, const float*, const float*, const float*, float*, float*, float*, int, int, int,
                     hipStream_t);
int se_train_bwd_f32(const float*, const float*, const float*, const float*, const float*, const float*, float*, float*, float*,
                     float*, float*, float*, float*, int, int, int, hipStream_t);
int dwconv_bwd_data_f32(const float*, const float*, int, int, int, int, int, int, int, int, int, int, float*, hipStream_t);
int dwconv_bwd_weight_f32(const float*, const float*, int, int, int, int, int, int, int, int, int, int, float*, float*, hipStream_t);
long long dwconv_bwd_weight_ws_floats(int, int, int, int, int);
int transpose_f32(const float*, int, int, int, float*, hipStream_t);
int im2col_f32(const float*, int, int, int, int, int, int, int, int, int, int, int, float*, hipStream_t);
int conv_wgrad_gather_f32(const float*, int, const float*, int, int, int, int, int, int, int, int, int, int, int, int, int, float*, float*,
                          hipStream_t);
long long conv_wgrad_gather_ws_floats(int, int, int, int, int);

static inline long long align4(long long n) { return (n + 3) & ~3LL; }

#define ICAP_TRY(expr)            \
    do {                          \
        const int rc_ = (expr);   \
        if (rc_ != ICAP_OK) return rc_; \
    } while (0)

}  // namespace icap

using namespace icap;

extern "C" {

int icap_abi_version(void) { return ICAP_ABI_VERSION; }

#ifndef ICAP_HEADER_HASH
#define ICAP_HEADER_HASH 0u
#endif
unsigned icap_header_hash(void) { return ICAP_HEADER_HASH; }
const char* icap_last_error(void) { return g_err; }

int icap_nchw_to_nhwc_pad_f32(const float* src, float* dst, int B, int C, int H, int W, int Cp, void* stream) {
    return nchw_to_nhwc_pad_f32(src, dst, B, C, H, W, Cp, (hipStream_t)stream);
}

int icap_conv2d_f32(const float* x, const float* w, const float* scale, const float* bias, const float* residual,
                    const float* gate, float* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                    int pad_t, int pad_l, int Ho, int Wo, int act, void* stream) {
    return conv_igemm_f32(x, w, scale, bias, residual, gate, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo,
                          act, (hipStream_t)stream);
}

int icap_conv2d_tune_f32(const float* x, const float* w, const float* scale, const float* bias, const float* residual,
                         const float* gate, float* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                         int pad_t, int pad_l, int Ho, int Wo, int act, void* stream) {
    return conv_igemm_tune_f32(x, w, scale, bias, residual, gate, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho,
                               Wo, act, (hipStream_t)stream);
}

int icap_conv2d_tuned_config(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int Ho, int Wo) {
    return conv_igemm_tuned_cfg(B, H, W, Cin, Cout, KH, KW, stride, Ho, Wo);
}

int icap_conv2d_num_configs(void) { return conv_igemm_num_cfgs(); }
int icap_conv2d_use_tuned(int flag) { return conv_igemm_use_tuned(flag); }

int icap_conv2d_cfg_f32(const float* x, const float* w, const float* scale, const float* bias, const float* residual,
                        const float* gate, float* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                        int pad_t, int pad_l, int Ho, int Wo, int act, int cfg, void* stream) {
    return conv_igemm_cfg_f32(x, w, scale, bias, residual, gate, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo,
                              act, cfg, (hipStream_t)stream);
}

int icap_linear_argmax_f32(const float* x, const float* w, const float* scale, const float* bias, float* y,
                           float* part_val, int* part_idx, int M, int K, int N, void* stream) {
    return linear_argmax_f32(x, w, scale, bias, y, part_val, part_idx, M, K, N, (hipStream_t)stream);
}

int icap_linear_argmax_parts(int N) { return linear_argmax_parts(N); }

int icap_next_token_f32(const float* part_val, const int* part_idx, int nparts, long long* ids, int ids_stride,
                        const float* table, const float* add, const float* gamma, const float* beta, float eps, float* out,
                        int ldo, int rows, int E, void* stream) {
    return next_token_f32(part_val, part_idx, nparts, ids, ids_stride, table, add, gamma, beta, eps, out, ldo, rows, E,
                          (hipStream_t)stream);
}

int icap_linear_bf16_f32(const void* x, int x_bf16, const unsigned short* w_bf16, const float* scale, const float* bias,
                         const float* residual, const float* gate, int hw, float* y, int M, int K, int N, int act, int cfg,
                         void* stream) {
    return linear_bf16_f32(x, x_bf16, w_bf16, scale, bias, residual, gate, hw, y, M, K, N, act, cfg, (hipStream_t)stream);
}

int icap_linear_bf16_argmax_f32(const float* x, const unsigned short* w_bf16, const float* scale, const float* bias, float* y,
                                float* part_val, int* part_idx, int M, int K, int N, void* stream) {
    return linear_bf16_argmax_f32(x, w_bf16, scale, bias, y, part_val, part_idx, M, K, N, (hipStream_t)stream);
}

int icap_linear_bf16_argmax_parts(int N) { return linear_bf16_argmax_parts(N); }

int icap_split3_f32(const float* src, unsigned short* dst, long long n, void* stream) { return split3_f32(src, dst, n, (hipStream_t)stream); }

int icap_linear_f32x3_f32(const void* x, int x_split, const unsigned short* w3, const float* scale, const float* bias,
                          const float* residual, const float* gate, int hw, float* y, int M, int K, int N, int act, int cfg,
                          void* stream) {
    return linear_f32x3_f32((const float*)x, x_split, w3, scale, bias, residual, gate, hw, y, M, K, N, act, cfg, (hipStream_t)stream);
}

int icap_linear_f32x3_argmax_f32(const void* x, int x_split, const unsigned short* w3, const float* scale, const float* bias,
                                 float* y, float* part_val, int* part_idx, int M, int K, int N, void* stream) {
    return linear_f32x3_argmax_f32((const float*)x, x_split, w3, scale, bias, y, part_val, part_idx, M, K, N, (hipStream_t)stream);
}

int icap_linear_x3dl_f32(const unsigned short* x3, const unsigned short* w3, const float* scale, const float* bias, const float* residual,
                         float* y, unsigned short* y3, int M, int K, int N, int act, int cfg, void* stream) {
    return linear_x3dl_f32(x3, w3, scale, bias, residual, y, y3, M, K, N, act, cfg, (hipStream_t)stream);
}

int icap_linear_x3dl_argmax_f32(const unsigned short* x3, const unsigned short* w3, const float* scale, const float* bias, float* y,
                                float* part_val, int* part_idx, int M, int K, int N, void* stream) {
    return linear_x3dl_argmax_f32(x3, w3, scale, bias, y, part_val, part_idx, M, K, N, (hipStream_t)stream);
}

int icap_to_bf16_f32(const float* src, unsigned short* dst, long long n, void* stream) {
    return to_bf16_f32(src, dst, n, (hipStream_t)stream);
}

int icap_topk_sample_f32(const float* logits, int ld, int rows, int V, int k, unsigned long long seed, unsigned long long offset,
                         float* probs, int* idx, long long* ids, int ids_stride, void* stream) {
    return topk_sample_f32(logits, ld, rows, V, k, seed, offset, probs, idx, ids, ids_stride, (hipStream_t)stream);
}

int icap_dwconv2d_f32(const float* x, const float* w, const float* scale, const float* bias, float* y, float* partial,
                      int B, int H, int W, int C, int KS, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                      void* stream) {
    return dwconv2d_f32(x, w, scale, bias, y, partial, B, H, W, C, KS, stride, pad_t, pad_l, Ho, Wo, act, (hipStream_t)stream);
}

int icap_dwconv_chunks(int Ho, int Wo, int C, int KS, int stride) { return dwconv_chunks(Ho, Wo, C, KS, stride); }

int icap_expand_dwconv_f32(const float* x, const float* w_exp, const float* exp_scale, const float* exp_bias, const float* w_dw,
                           const float* dw_scale, const float* dw_bias, float* y, float* partial, int B, int H, int W, int Cin,
                           int Cexp, int KS, int stride, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
    return expand_dwconv_f32(x, w_exp, exp_scale, exp_bias, w_dw, dw_scale, dw_bias, y, partial, B, H, W, Cin, Cexp, KS, stride,
                             pad_t, pad_l, Ho, Wo, 0, (hipStream_t)stream);
}

int icap_expand_dwconv_bf16_f32(const float* x, const float* w_exp, const float* exp_scale, const float* exp_bias, const float* w_dw,
                                const float* dw_scale, const float* dw_bias, void* y, int y_bf16, float* partial, int B, int H, int W,
                                int Cin, int Cexp, int KS, int stride, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
    return expand_dwconv_f32(x, w_exp, exp_scale, exp_bias, w_dw, dw_scale, dw_bias, (float*)y, partial, B, H, W, Cin, Cexp, KS,
                             stride, pad_t, pad_l, Ho, Wo, 1 | (y_bf16 ? 2 : 0), (hipStream_t)stream);
}

int icap_expand_dwconv_f32x3_f32(const float* x, const float* w_exp, const float* exp_scale, const float* exp_bias, const float* w_dw,
                                 const float* dw_scale, const float* dw_bias, float* y, float* partial, int B, int H, int W, int Cin,
                                 int Cexp, int KS, int stride, int pad_t, int pad_l, int Ho, int Wo, void* stream) {
    return expand_dwconv_f32(x, w_exp, exp_scale, exp_bias, w_dw, dw_scale, dw_bias, y, partial, B, H, W, Cin, Cexp, KS, stride,
                             pad_t, pad_l, Ho, Wo, 4, (hipStream_t)stream);
}

int icap_stem_dwconv_f32(const float* x, const float* w_stem, const float* stem_scale, const float* stem_bias, const float* w_dw,
                         const float* dw_scale, const float* dw_bias, float* y, float* partial, int B, int img_h, int img_w,
                         int Hs, int Ws, int C, int stem_pad_t, int stem_pad_l, int pad_t, int pad_l, int Ho, int Wo,
                         void* stream) {
    return stem_dwconv_f32(x, w_stem, stem_scale, stem_bias, w_dw, dw_scale, dw_bias, y, partial, B, img_h, img_w, Hs, Ws, C,
                           stem_pad_t, stem_pad_l, pad_t, pad_l, Ho, Wo, 0, (hipStream_t)stream);
}

int icap_stem_dwconv_nchw_f32(const float* x, const float* w_stem, const float* stem_scale, const float* stem_bias, const float* w_dw,
                              const float* dw_scale, const float* dw_bias, float* y, float* partial, int B, int img_h, int img_w,
                              int Hs, int Ws, int C, int stem_pad_t, int stem_pad_l, int pad_t, int pad_l, int Ho, int Wo,
                              void* stream) {
    return stem_dwconv_f32(x, w_stem, stem_scale, stem_bias, w_dw, dw_scale, dw_bias, y, partial, B, img_h, img_w, Hs, Ws, C,
                           stem_pad_t, stem_pad_l, pad_t, pad_l, Ho, Wo, 8, (hipStream_t)stream);
}

int icap_stem_dwconv_valu_supported(int C) { return stem_dwconv_valu_supported(C); }
int icap_stem_dwconv_valu_f32(const float* x, const float* w_stem_k, const float* stem_scale, const float* stem_bias, const float* w_dw,
                              const float* dw_scale, const float* dw_bias, float* y, float* partial, int B, int img_h, int img_w,
                              int Hs, int Ws, int C, int stem_pad_t, int stem_pad_l, int pad_t, int pad_l, int Ho, int Wo,
                              void* stream) {
    return stem_dwconv_valu_f32(x, w_stem_k, stem_scale, stem_bias, w_dw, dw_scale, dw_bias, y, partial, B, img_h, img_w, Hs, Ws, C,
                                stem_pad_t, stem_pad_l, pad_t, pad_l, Ho, Wo, (hipStream_t)stream);
}

int icap_stem_dwconv_bf16_f32(const float* x, const float* w_stem, const float* stem_scale, const float* stem_bias, const float* w_dw,
                              const float* dw_scale, const float* dw_bias, void* y, int y_bf16, float* partial, int B, int img_h,
                              int img_w, int Hs, int Ws, int C, int stem_pad_t, int stem_pad_l, int pad_t, int pad_l, int Ho, int Wo,
                              void* stream) {
    return stem_dwconv_f32(x, w_stem, stem_scale, stem_bias, w_dw, dw_scale, dw_bias, (float*)y, partial, B, img_h, img_w, Hs, Ws, C,
                           stem_pad_t, stem_pad_l, pad_t, pad_l, Ho, Wo, 1 | (y_bf16 ? 2 : 0), (hipStream_t)stream);
}

int icap_stem_dwconv_chunks(int Ho, int Wo, int C) { return stem_dwconv_chunks(Ho, Wo, C); }

int icap_expand_dwconv_chunks(int Ho, int Wo, int Cin, int Cexp, int KS, int stride) {
    return expand_dwconv_chunks(Ho, Wo, Cin, Cexp, KS, stride);
}

int icap_expand_dwconv_whole_image(int H, int W, int Cin, int Cexp, int KS, int stride) {
    return expand_dwconv_whole_image(H, W, Cin, Cexp, KS, stride);
}

long long icap_expand_dwconv_img_x3_we_bytes(int H, int W, int Cin, int Cexp, int KS) {
    return expand_dwconv_img_x3_we_bytes(H, W, Cin, Cexp, KS);
}

int icap_expand_dwconv_img_x3_pack(const float* w_exp, void* w_packed, int H, int W, int Cin, int Cexp, int KS, void* stream) {
    return expand_dwconv_img_x3_pack(w_exp, w_packed, H, W, Cin, Cexp, KS, (hipStream_t)stream);
}

int icap_expand_dwconv_img_x3_f32(const float* x, const void* w_packed, const float* exp_scale, const float* exp_bias, const float* w_dw,
                                  const float* dw_scale, const float* dw_bias, float* y, float* pooled, int B, int H, int W, int Cin,
                                  int Cexp, int KS, int pad_t, int pad_l, void* stream) {
    return expand_dwconv_img_x3_f32(x, w_packed, exp_scale, exp_bias, w_dw, dw_scale, dw_bias, y, pooled, B, H, W, Cin, Cexp, KS, pad_t,
                                    pad_l, (hipStream_t)stream);
}

int icap_expand_dwconv_img_x3_s2_supported(int H, int W, int Cin, int Cexp, int KS, int pad_t, int pad_l) {
    return expand_dwconv_img_x3_s2_supported(H, W, Cin, Cexp, KS, pad_t, pad_l);
}

int icap_expand_dwconv_img_x3_s2_f32(const float* x, const void* w_packed, const float* exp_scale, const float* exp_bias, const float* w_dw,
                                     const float* dw_scale, const float* dw_bias, float* y, float* pooled, int B, int H, int W, int Cin,
                                     int Cexp, int KS, int pad_t, int pad_l, void* stream) {
    return expand_dwconv_img_x3_s2_f32(x, w_packed, exp_scale, exp_bias, w_dw, dw_scale, dw_bias, y, pooled, B, H, W, Cin, Cexp, KS, pad_t,
                                       pad_l, (hipStream_t)stream);
}

int icap_se_gate_f32(const float* partial, int nchunks, float inv_hw, const float* wr, const float* br, const float* we,
                     const float* be, float* gate, int B, int C, int S, void* stream) {
    return se_gate_f32(partial, nchunks, inv_hw, wr, br, we, be, gate, B, C, S, (hipStream_t)stream);
}

int icap_maxpool2d_f32(const float* x, float* y, int B, int H, int W, int C, int KS, int stride, int pad, int Ho, int Wo,
                       void* stream) {
    return maxpool2d_f32(x, y, B, H, W, C, KS, stride, pad, Ho, Wo, (hipStream_t)stream);
}

int icap_adaptive_avgpool2d_f32(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, void* stream) {
    return adaptive_avgpool2d_f32(x, y, B, H, W, C, OH, OW, (hipStream_t)stream);
}

int icap_embedding_f32(const float* table, const long long* idx, const float* add, float* out, int rows, int E, int V,
                       int* err_flag, void* stream) {
    return embedding_f32(table, idx, add, out, E, rows, E, V, err_flag, (hipStream_t)stream);
}

int icap_layernorm_f32(const float* x, const float* res, int res_rows, const float* gamma, const float* beta, float* y,
                       int rows, int E, float eps, float pre_scale, void* stream) {
    return layernorm_f32(x, res, res_rows, gamma, beta, y, rows, E, eps, pre_scale, (hipStream_t)stream);
}

int icap_layernorm_split3_f32(const float* x, const float* res, int res_rows, const float* gamma, const float* beta, float* y,
                              unsigned short* y3, int rows, int E, float eps, float pre_scale, void* stream) {
    return layernorm_split3_f32(x, res, res_rows, gamma, beta, y, y3, rows, E, eps, pre_scale, (hipStream_t)stream);
}

int icap_argmax_rows_f32(const float* logits, int ld, long long* idx, int idx_stride, int rows, int V, float* top2gap,
                         void* stream) {
    return argmax_rows_f32(logits, ld, idx, idx_stride, rows, V, top2gap, (hipStream_t)stream);
}

int icap_mha_core_f32(const float* q, const float* k, const float* v, float* out, int Lq, int Lk, int N, int Nk, int E,
                      int heads, void* stream) {
    return mha_core_f32(q, k, v, out, Lq, Lk, N, Nk, E, heads, (hipStream_t)stream);
}

int icap_bahdanau_attention_f32(const float* u, const float* w, const float* a_w, float a_b, const float* feat,
                                float* alpha, float* context, int B, int A, int E, void* stream) {
    return bahdanau_attention_f32(u, w, a_w, a_b, feat, alpha, context, B, A, E, (hipStream_t)stream);
}

int icap_lstm_pointwise_f32(const float* gates, const float* c_in, float* h_out, float* c_out, int B, int H, void* stream) {
    return lstm_pointwise_f32(gates, c_in, h_out, c_out, B, H, (hipStream_t)stream);
}

int icap_gru_pointwise_f32(const float* gi, const float* gh, const float* h_in, float* h_out, int B, int H, void* stream) {
    return gru_pointwise_f32(gi, gh, h_in, h_out, B, H, (hipStream_t)stream);
}

int icap_gemv_f32(const float* W, const float* scale, const float* bias, const float* x, int ldx, float* y, int ldy,
                  int NB, int N, int K, int act, float* part_val, int* part_idx, void* stream) {
    return gemv_f32(W, scale, bias, x, ldx, y, ldy, NB, N, K, act, part_val, part_idx, (hipStream_t)stream);
}

int icap_gemv_blocks(int N) { return gemv_blocks(N); }

int icap_argmax_finalize_f32(const float* part_val, const int* part_idx, int nparts, long long* ids, int ids_stride,
                             const float* table, const float* add, float* next_x, int ldx, int E, int NB, void* stream) {
    return argmax_finalize_f32(part_val, part_idx, nparts, ids, ids_stride, table, add, next_x, ldx, E, NB, (hipStream_t)stream);
}

int icap_lstm_cell_f32(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* x, int ldx,
                       const float* h_in, const float* c_in, float* h_out, float* c_out, int NB, int I, int H, void* stream) {
    return lstm_cell_f32(w_ih, w_hh, b_ih, b_hh, x, ldx, h_in, c_in, h_out, c_out, NB, I, H, (hipStream_t)stream);
}

int icap_gru_cell_f32(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* x, int ldx,
                      const float* h_in, float* h_out, int NB, int I, int H, void* stream) {
    return gru_cell_f32(w_ih, w_hh, b_ih, b_hh, x, ldx, h_in, h_out, NB, I, H, (hipStream_t)stream);
}

// ------------------------------------------------------------------ greedy decode drivers
// A step is 2-3 launches: the first kernel of step t+1 reduces the arg-max partials the
// vocabulary projection of step t left behind, stores token t and gathers its embedding row
// (StepInput), so no kernel exists just to pick the token; one finalize launch after the
// loop stores the last token.

static StepInput token_input(const float* pv, const int* pi, int P, const long long* tok_in, long long* ids_out,
                             int ids_stride, const float* table, int E, int rows) {
    StepInput s;
    s.table = table;
    s.E = E;
    s.table_rows = rows;
    if (tok_in) {
        s.tok_in = tok_in;
    } else {
        s.part_val = pv;
        s.part_idx = pi;
        s.nparts = P;
        s.ids_out = ids_out;
        s.ids_stride = ids_stride;
    }
    return s;
}

// ------------------------------------------------------------------ greedy: seq2seq LSTM / GRU
// workspace (floats): h [2][layers][NB][H] | c [2][layers][NB][H] | part_val [NB][P] | part_idx [NB][P]
long long icap_greedy_rnn_ws_floats(int NB, int layers, int H, int I, int V) {
    const long long P = gemv_blocks(V);
    (void)I;
    return 4LL * layers * align4((long long)NB * H) + 2 * align4((long long)NB * P);
}

// ---- the same driver for callers that cannot allocate: its workspace is a fixed array in the library's device image.
// Every GPU holds its own copy of the array at its own address, looked up once per device (hipGetSymbolAddress with that
// device current) and kept; host code only ever uses the looked-up pointer, never the array's name.
static __device__ __align__(256) float scratch[ICAP_GREEDY_RNN_STATIC_WS_FLOATS];
static float* g_scratch_ptr[64];
static std::mutex g_scratch_mu;

static int static_scratch(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) { set_error("greedy_rnn_static: no current device (%s)", hipGetErrorString(e)); return ICAP_ERR_HIP_BASE + (int)e; }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) { set_error("greedy_rnn_static: hipGetSymbolAddress: %s", hipGetErrorString(e)); return ICAP_ERR_HIP_BASE + (int)e; }
        g_scratch_ptr[dev] = static_cast<float*>(p);
    }
    *out = g_scratch_ptr[dev];
    return ICAP_OK;
}

int icap_greedy_rnn_static_ws_floats(void) { return ICAP_GREEDY_RNN_STATIC_WS_FLOATS; }

int icap_greedy_rnn_static_f32(const icap_rnn_weights* w, const float* x0, long long* ids, float* logits_out, int NB, int steps,
                               void* stream) {
    if (!w || NB <= 0 || w->layers <= 0 || w->H <= 0 || w->V <= 0) { set_error("greedy_rnn_static: bad arguments"); return ICAP_ERR_ARG; }
    const long long need = icap_greedy_rnn_ws_floats(NB, w->layers, w->H, w->I, w->V);
    if (need > ICAP_GREEDY_RNN_STATIC_WS_FLOATS) {
        set_error("greedy_rnn_static: NB=%d layers=%d H=%d V=%d needs %lld workspace floats, the built-in buffer holds %d; "
                  "use icap_greedy_rnn_f32 with a caller workspace", NB, w->layers, w->H, w->V, need, ICAP_GREEDY_RNN_STATIC_WS_FLOATS);
        return ICAP_ERR_LIMIT;
    }
    float* ws = nullptr;
    ICAP_TRY(static_scratch(&ws));
    // the driver clears the recurrent state itself (memset of h and c on the stream) and writes the arg-max partials
    // before reading them, so nothing a previous call left in the buffer is ever read
    return icap_greedy_rnn_f32(w, x0, ids, logits_out, ws, NB, steps, stream);
}

static int greedy_rnn_impl(const icap_rnn_weights* w, const float* x0, long long* ids, float* logits_out, float* ws, int NB,
                           int steps, void* stream, bool w16) {
    if (!w || !x0 || !ids || !ws || NB <= 0 || steps <= 0) { set_error("greedy_rnn: bad arguments"); return ICAP_ERR_ARG; }
    if (w->layers <= 0 || w->layers > 4 || w->H <= 0 || w->I <= 0 || w->V <= 0 || !w->fc_w || !w->fc_b || !w->in_table) {
        set_error("greedy_rnn: bad weight descriptor");
        return ICAP_ERR_ARG;
    }
    if (NB > 8) { set_error("greedy_rnn: NB=%d exceeds 8 captions per call", NB); return ICAP_ERR_LIMIT; }
    hipStream_t st = (hipStream_t)stream;
    const int L = w->layers, H = w->H, I = w->I, V = w->V;
    const int P = gemv_blocks(V);
    const long long hsz = align4((long long)NB * H);
    float* hbuf = ws;
    float* cbuf = hbuf + 2LL * L * hsz;
    float* pv = cbuf + 2LL * L * hsz;
    int* pi = reinterpret_cast<int*>(pv + align4((long long)NB * P));
    // hidden=None in the reference means zero state (seq2seqbasic.py:74)
    hipError_t e = hipMemsetAsync(hbuf, 0, sizeof(float) * 4LL * L * hsz, st);
    if (e != hipSuccess) { set_error("greedy_rnn: memset: %s", hipGetErrorString(e)); return ICAP_ERR_HIP_BASE + (int)e; }
    int cur = 0;
    for (int t = 0; t < steps; ++t) {
        // step 0 consumes the image feature; later steps feature_fc(embed(previous token)) from in_table
        StepInput in = t == 0 ? plain_input(x0, I) : token_input(pv, pi, P, nullptr, ids + (t - 1), steps, w->in_table, I, V);
        int in_w = I;
        for (int l = 0; l < L; ++l) {
            const float* h_in = hbuf + (2LL * l + cur) * hsz;
            float* h_out = hbuf + (2LL * l + (cur ^ 1)) * hsz;
            const float* c_in = cbuf + (2LL * l + cur) * hsz;
            float* c_out = cbuf + (2LL * l + (cur ^ 1)) * hsz;
            ICAP_TRY(rnn_cell_step_f32(w->is_lstm != 0, w->w_ih[l], w->w_hh[l], w->b_ih[l], w->b_hh[l], in, h_in, c_in, h_out,
                                       c_out, NB, in_w, H, st, w16));
            in = plain_input(h_out, H);
            in_w = H;
        }
        float* lo = logits_out ? logits_out + (size_t)t * NB * V : nullptr;
        ICAP_TRY(gemv_step_f32(w->fc_w, nullptr, w->fc_b, in, lo, V, NB, V, H, ACT_NONE, pv, pi, st, w16));
        cur ^= 1;
    }
    return argmax_finalize_f32(pv, pi, P, ids + (steps - 1), steps, nullptr, nullptr, nullptr, 0, 0, NB, st);
}

int icap_greedy_rnn_f32(const icap_rnn_weights* w, const float* x0, long long* ids, float* logits_out, float* ws, int NB,
                        int steps, void* stream) {
    return greedy_rnn_impl(w, x0, ids, logits_out, ws, NB, steps, stream, false);
}

int icap_greedy_rnn_w16_f32(const icap_rnn_weights* w, const float* x0, long long* ids, float* logits_out, float* ws, int NB,
                            int steps, void* stream) {
    return greedy_rnn_impl(w, x0, ids, logits_out, ws, NB, steps, stream, true);
}

// ------------------------------------------------------------------ greedy: word / char transformer
// workspace: x [NB][E] | y [NB][E] | h [NB][FF] | part_val | part_idx
long long icap_greedy_tfm_ws_floats(int NB, int E, int FF, int V) {
    const long long P = gemv_blocks(V);
    return 2 * align4((long long)NB * E) + align4((long long)NB * FF) + 2 * align4((long long)NB * P);
}

static int greedy_tfm_impl(const icap_tfm_weights* w, const float* ca, const long long* start_ids, long long* ids,
                           float* logits_out, float* ws, int NB, int steps, void* stream, bool w16) {
    if (!w || !ca || !start_ids || !ids || !ws || NB <= 0 || steps <= 0) { set_error("greedy_tfm: bad arguments"); return ICAP_ERR_ARG; }
    if (w->E <= 0 || w->FF <= 0 || w->V <= 0 || !w->emb || !w->ln2_g || !w->ln2_b || !w->ln3_g || !w->ln3_b || !w->fc1_w ||
        !w->fc1_b || !w->fc2_w || !w->fc2_b || !w->out_w || !w->out_b) {
        set_error("greedy_tfm: bad weight descriptor");
        return ICAP_ERR_ARG;
    }
    if (NB > 8) { set_error("greedy_tfm: NB=%d exceeds 8 captions per call", NB); return ICAP_ERR_LIMIT; }
    hipStream_t st = (hipStream_t)stream;
    const int E = w->E, FF = w->FF, V = w->V, P = gemv_blocks(V);
    const long long esz = align4((long long)NB * E);
    float* x = ws;
    float* y = x + esz;
    float* h = y + esz;
    float* pv = h + align4((long long)NB * FF);
    int* pi = reinterpret_cast<int*>(pv + align4((long long)NB * P));
    for (int t = 0; t < steps; ++t) {
        // x = LN2(embedding(token) + cross-attention row); h = relu(fc1 x)
        StepInput in1 = token_input(pv, pi, P, t == 0 ? start_ids : nullptr, ids + (t - 1), steps, w->emb, E, V);
        in1.add = ca;
        in1.ld_add = E;
        in1.ln_g = w->ln2_g;
        in1.ln_b = w->ln2_b;
        in1.x_out = x;
        in1.ld_xout = E;
        ICAP_TRY(gemv_step_f32(w->fc1_w, nullptr, w->fc1_b, in1, h, FF, NB, FF, E, ACT_RELU, nullptr, nullptr, st, w16));
        ICAP_TRY(gemv_step_f32(w->fc2_w, nullptr, w->fc2_b, plain_input(h, FF), y, E, NB, E, FF, ACT_NONE, nullptr, nullptr, st, w16));
        // logits = out(LN3(x + y))
        StepInput in3 = plain_input(y, E);
        in3.add = x;
        in3.ld_add = E;
        in3.ln_g = w->ln3_g;
        in3.ln_b = w->ln3_b;
        float* lo = logits_out ? logits_out + (size_t)t * NB * V : nullptr;
        ICAP_TRY(gemv_step_f32(w->out_w, nullptr, w->out_b, in3, lo, V, NB, V, E, ACT_NONE, pv, pi, st, w16));
    }
    return argmax_finalize_f32(pv, pi, P, ids + (steps - 1), steps, nullptr, nullptr, nullptr, 0, 0, NB, st);
}

int icap_greedy_tfm_f32(const icap_tfm_weights* w, const float* ca, const long long* start_ids, long long* ids,
                        float* logits_out, float* ws, int NB, int steps, void* stream) {
    return greedy_tfm_impl(w, ca, start_ids, ids, logits_out, ws, NB, steps, stream, false);
}

int icap_greedy_tfm_w16_f32(const icap_tfm_weights* w, const float* ca, const long long* start_ids, long long* ids,
                            float* logits_out, float* ws, int NB, int steps, void* stream) {
    return greedy_tfm_impl(w, ca, start_ids, ids, logits_out, ws, NB, steps, stream, true);
}

int icap_table_walk(const long long* next, const long long* start, long long* ids, const float* logits_tab, int ldl,
                    float* logits_out, int V, int B, int steps, void* stream) {
    return table_walk(next, start, ids, logits_tab, ldl, logits_out, V, B, steps, (hipStream_t)stream);
}

int icap_linear_thin_supported(int K, int N) { return linear_thin_supported(K, N); }
int icap_linear_f32x3_tall_supported(int K, int N, int hw, int gated) { return linear_f32x3_tall_supported(K, N, hw, gated); }
long long icap_linear_f32x3_splitk_ws_floats(int M, int N, int ksplit) { return linear_f32x3_splitk_ws_floats(M, N, ksplit); }
int icap_linear_f32x3_splitk_f32(const float* x, const unsigned short* w3, const float* scale, const float* bias, const float* residual,
                                 const float* gate, int hw, float* y, float* ws, long long ws_floats, int M, int K, int N, int act, int cfg,
                                 int ksplit, void* stream) {
    return linear_f32x3_splitk_f32(x, w3, scale, bias, residual, gate, hw, y, ws, ws_floats, M, K, N, act, cfg, ksplit, (hipStream_t)stream);
}
int icap_linear_thin_f32(const float* x, const float* w, const float* scale, const float* bias, const float* residual, const float* gate,
                         int hw, float* y, long long M, int K, int N, void* stream) {
    return linear_thin_f32(x, w, scale, bias, residual, gate, hw, y, M, K, N, (hipStream_t)stream);
}

int icap_greedy_rnn2_persist_max_nb(int is_lstm, int H, int V) { return greedy_rnn2_persist_max_nb(is_lstm, H, V); }
long long icap_greedy_rnn2_persist_ws_bytes(int NB, int is_lstm, int H, int V) { return greedy_rnn2_persist_ws_bytes(NB, is_lstm, H, V); }
int icap_greedy_rnn2_persist_f32(int is_lstm, const float* g_tab0, const float* g_first, const float* w_hh0, const float* hh0_bias,
                                 const float* w_ih1, const float* ih1_bias, const float* w_hh1, const float* hh1_bias, const float* fc_w,
                                 const float* fc_b, long long* ids, float* logits_out, void* ws, int NB, int H, int V, int steps,
                                 void* stream) {
    return greedy_rnn2_persist_f32(is_lstm, g_tab0, g_first, w_hh0, hh0_bias, w_ih1, ih1_bias, w_hh1, hh1_bias, fc_w, fc_b, ids, logits_out,
                                   ws, NB, H, V, steps, (hipStream_t)stream);
}

int icap_greedy_bah_persist_max_nb(int H, int V) { return greedy_bah_persist_max_nb(H, V); }
long long icap_greedy_bah_persist_ws_bytes(int NB, int H, int V) { return greedy_bah_persist_ws_bytes(NB, H, V); }
int icap_greedy_bah_persist_f32(const float* g_tab, const float* g_ctx, const float* w_hh, const float* fc_w, const float* fc_scale,
                                const float* fc_bias, const long long* start_ids, long long* ids, float* logits_out, void* ws,
                                int NB, int H, int V, int steps, void* stream) {
    return greedy_bah_persist_f32(g_tab, g_ctx, w_hh, fc_w, fc_scale, fc_bias, start_ids, ids, logits_out, ws, NB, H, V, steps,
                                  (hipStream_t)stream);
}
int icap_greedy_tfm_persist_max_nb(int E, int FF, int V) { return greedy_tfm_persist_max_nb(E, FF, V); }
long long icap_greedy_tfm_persist_ws_bytes(int NB, int E, int FF, int V) { return greedy_tfm_persist_ws_bytes(NB, E, FF, V); }
int icap_greedy_tfm_persist_f32(const icap_tfm_weights* w, const float* ca, const long long* start_ids, long long* ids,
                                float* logits_out, void* ws, int NB, int steps, void* stream) {
    return greedy_tfm_persist_f32(w, ca, start_ids, ids, logits_out, ws, NB, steps, (hipStream_t)stream);
}

// ------------------------------------------------------------------ greedy: Bahdanau LSTMCell decoder
// workspace: h [2][NB][H] | c [2][NB][H] | part_val | part_idx
long long icap_greedy_bah_ws_floats(int NB, int E, int ENC, int H, int V) {
    const long long P = gemv_blocks(V);
    (void)E;
    (void)ENC;
    return 4 * align4((long long)NB * H) + 2 * align4((long long)NB * P);
}

static int greedy_bah_impl(const icap_bah_weights* w, const float* context, int ENC, const long long* start_ids,
                           long long* ids, float* logits_out, float* ws, int NB, int steps, void* stream, bool w16) {
    if (!w || !context || !start_ids || !ids || !ws || NB <= 0 || steps <= 0 || ENC <= 0) { set_error("greedy_bah: bad arguments"); return ICAP_ERR_ARG; }
    if (w->E <= 0 || w->H <= 0 || w->V <= 0 || !w->emb || !w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh || !w->fc_w) {
        set_error("greedy_bah: bad weight descriptor");
        return ICAP_ERR_ARG;
    }
    if (NB > 8) { set_error("greedy_bah: NB=%d exceeds 8 captions per call", NB); return ICAP_ERR_LIMIT; }
    if ((w->E % 4) || (ENC % 4)) { set_error("greedy_bah: E and ENC must be multiples of 4"); return ICAP_ERR_ALIGN; }
    if (w16 && ((w->E + ENC) % 8)) { set_error("greedy_bah: bf16 weight rows need E + ENC to be a multiple of 8"); return ICAP_ERR_ALIGN; }
    hipStream_t st = (hipStream_t)stream;
    const int E = w->E, H = w->H, V = w->V, I = E + ENC, P = gemv_blocks(V);
    const long long hsz = align4((long long)NB * H);
    float* hbuf = ws;
    float* cbuf = hbuf + 2 * hsz;
    float* pv = cbuf + 2 * hsz;
    int* pi = reinterpret_cast<int*>(pv + align4((long long)NB * P));
    hipError_t e = hipMemsetAsync(hbuf, 0, sizeof(float) * 4 * hsz, st);  // init_hidden_state: zeros
    if (e != hipSuccess) { set_error("greedy_bah: memset: %s", hipGetErrorString(e)); return ICAP_ERR_HIP_BASE + (int)e; }
    int cur = 0;
    for (int t = 0; t < steps; ++t) {
        const float* h_in = hbuf + cur * hsz;
        float* h_out = hbuf + (cur ^ 1) * hsz;
        const float* c_in = cbuf + cur * hsz;
        float* c_out = cbuf + (cur ^ 1) * hsz;
        // x[b] = [ emb[token] | context[b] ]; the context half never changes
        StepInput in = token_input(pv, pi, P, t == 0 ? start_ids : nullptr, ids + (t - 1), steps, w->emb, E, V);
        in.tail = context;
        in.ld_tail = ENC;
        ICAP_TRY(rnn_cell_step_f32(true, w->w_ih, w->w_hh, w->b_ih, w->b_hh, in, h_in, c_in, h_out, c_out, NB, I, H, st, w16));
        float* lo = logits_out ? logits_out + (size_t)t * NB * V : nullptr;
        ICAP_TRY(gemv_step_f32(w->fc_w, w->fc_scale, w->fc_bias, plain_input(h_out, H), lo, V, NB, V, H, ACT_NONE, pv, pi, st, w16));
        cur ^= 1;
    }
    return argmax_finalize_f32(pv, pi, P, ids + (steps - 1), steps, nullptr, nullptr, nullptr, 0, 0, NB, st);
}

int icap_greedy_bah_f32(const icap_bah_weights* w, const float* context, int ENC, const long long* start_ids,
                        long long* ids, float* logits_out, float* ws, int NB, int steps, void* stream) {
    return greedy_bah_impl(w, context, ENC, start_ids, ids, logits_out, ws, NB, steps, stream, false);
}

int icap_greedy_bah_w16_f32(const icap_bah_weights* w, const float* context, int ENC, const long long* start_ids,
                            long long* ids, float* logits_out, float* ws, int NB, int steps, void* stream) {
    return greedy_bah_impl(w, context, ENC, start_ids, ids, logits_out, ws, NB, steps, stream, true);
}

// ------------------------------------------------------------------ greedy: fewer dependent launches per step (decode_fused.hip)
long long icap_greedy_bah_fused_ws_floats(int NB, int H, int V) { return greedy_bah_fused_ws_floats(NB, H, V); }

int icap_greedy_bah_fused_f32(const float* g_tab, const float* g_ctx, const float* w_hh, const float* fc_w, const float* fc_scale,
                              const float* fc_bias, const long long* start_ids, long long* ids, float* logits_out, float* ws,
                              int NB, int H, int V, int steps, void* stream) {
    return greedy_bah_fused_f32(g_tab, g_ctx, w_hh, fc_w, fc_scale, fc_bias, start_ids, ids, logits_out, ws, NB, H, V, steps,
                                (hipStream_t)stream);
}

long long icap_greedy_rnn2_fused_ws_floats(int NB, int is_lstm, int H, int V) { return greedy_rnn2_fused_ws_floats(NB, is_lstm ? 4 : 3, H, V); }

int icap_greedy_rnn2_fused_f32(int is_lstm, const float* g_tab0, const float* g_first, const float* w_hh0, const float* hh0_bias,
                               const float* w_ih1, const float* ih1_bias, const float* w_hh1, const float* hh1_bias,
                               const float* fc_w, const float* fc_b, long long* ids, float* logits_out, float* ws, int NB, int H,
                               int V, int steps, void* stream) {
    return greedy_rnn2_fused_f32(is_lstm, g_tab0, g_first, w_hh0, hh0_bias, w_ih1, ih1_bias, w_hh1, hh1_bias, fc_w, fc_b, ids,
                                 logits_out, ws, NB, H, V, steps, (hipStream_t)stream);
}

// ------------------------------------------------------------------ training step
int icap_ce_loss_fwd_f32(const float* logits, int ld, int group_rows, int group_stride, const long long* targets, int R,
                         int V, long long ignore_index, int mean_over_valid, float* row_loss, float* row_lse, float* out,
                         void* stream) {
    return ce_loss_fwd_f32(logits, ld, group_rows, group_stride, targets, R, V, ignore_index, mean_over_valid, row_loss, row_lse, out, (hipStream_t)stream);
}

int icap_ce_loss_bwd_f32(const float* logits, int ld, int group_rows, int group_stride, const long long* targets, int R,
                         int V, long long ignore_index, const float* row_lse, const float* meta, const float* grad_out,
                         float* dlogits, int ldd, void* stream) {
    return ce_loss_bwd_f32(logits, ld, group_rows, group_stride, targets, R, V, ignore_index, row_lse, meta, grad_out, dlogits, ldd, (hipStream_t)stream);
}

int icap_adamw_dev_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, const int* step_dev, int step_bias, float grad_scale, void* stream) {
    return adamw_dev_f32(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step_dev, step_bias, grad_scale, (hipStream_t)stream);
}

int icap_dropout_dev_f32(const float* x, float p, unsigned long long seed, unsigned long long offset, const unsigned long long* offset_base,
                         long long n, long long group, float* y, float* mask, void* stream) {
    return dropout_dev_f32(x, p, seed, offset, offset_base, n, group, y, mask, (hipStream_t)stream);
}

int icap_adamw_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float grad_scale, void* stream) {
    return adamw_f32(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, (hipStream_t)stream);
}

int icap_gemm_bwd_f32(const float* a, int lda, int a_kmajor, const float* b, int ldb, float* c, int ldc, int M, int N, int K,
                      int accumulate, float* ws, void* stream) {
    return gemm_bwd_f32(a, lda, a_kmajor, b, ldb, c, ldc, M, N, K, accumulate, ws, (hipStream_t)stream);
}
long long icap_gemm_bwd_ws_floats(int M, int N, int K, int accumulate) { return gemm_bwd_ws_floats(M, N, K, accumulate); }

int icap_col_reduce_f32(int mode, const float* a, int lda, const float* x, int ldx, const float* v1, const float* v2,
                        const float* v3, const float* v4, const float* r1, const float* r2, int act, int M, int C, float scale,
                        int accumulate, float* out1, float* out2, float* ws, void* stream) {
    return col_reduce_f32(mode, a, lda, x, ldx, v1, v2, v3, v4, r1, r2, act, M, C, scale, accumulate, out1, out2, ws, (hipStream_t)stream);
}
long long icap_col_reduce_ws_floats(int M, int C) { return col_reduce_ws_floats(M, C); }

int icap_bn_train_stats_f32(const float* x, int ldx, const float* gamma, const float* beta, float eps, float momentum, int M, int C,
                            float* mean, float* rstd, float* s, float* b, float* run_mean, float* run_var, float* ws, void* stream) {
    return bn_train_stats_f32(x, ldx, gamma, beta, eps, momentum, M, C, mean, rstd, s, b, run_mean, run_var, ws, (hipStream_t)stream);
}

long long icap_conv2d_stats_max_parts(long long M) { return conv_stats_max_parts(M); }

int icap_conv2d_stats_f32(const float* x, const float* w, float* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                          int pad_t, int pad_l, int Ho, int Wo, float* stat_part, long long part_floats, int* info, void* stream) {
    return conv_igemm_stats_f32(x, w, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, stat_part, part_floats, info,
                                (hipStream_t)stream);
}

int icap_linear_f32x3_stats_f32(const float* x, const unsigned short* w3, float* y, int M, int K, int N, int cfg, float* stat_part,
                                long long part_floats, int* info, void* stream) {
    return linear_f32x3_stats_f32(x, w3, y, M, K, N, cfg, stat_part, part_floats, info, (hipStream_t)stream);
}

int icap_dwconv2d_stats_f32(const float* x, const float* w, float* y, int B, int H, int W, int C, int KS, int stride, int pad_t,
                            int pad_l, int Ho, int Wo, float* stat_part, float* stat_cnt, void* stream) {
    return dwconv2d_stats_f32(x, w, y, B, H, W, C, KS, stride, pad_t, pad_l, Ho, Wo, stat_part, stat_cnt, (hipStream_t)stream);
}

long long icap_bn_stats_from_parts_ws_floats(int P, int C) { return bn_stats_from_parts_ws_floats(P, C); }

int icap_bn_stats_from_parts_f32(const float* part, const float* cnt, int P, int rows_per_part, int M, int C, const float* gamma,
                                 const float* beta, float eps, float momentum, float* mean, float* rstd, float* s, float* b,
                                 float* run_mean, float* run_var, float* ws, void* stream) {
    return bn_stats_from_parts_f32(part, cnt, P, rows_per_part, M, C, gamma, beta, eps, momentum, mean, rstd, s, b, run_mean, run_var, ws,
                                   (hipStream_t)stream);
}

int icap_bn_finalize_moments_f32(const float* shift, const float* s1, const float* s2, const float* gamma, const float* beta,
                                 float eps, float momentum, int M, int C, float* mean, float* rstd, float* s, float* b,
                                 float* run_mean, float* run_var, void* stream) {
    return bn_finalize_moments_f32(shift, s1, s2, gamma, beta, eps, momentum, M, C, mean, rstd, s, b, run_mean, run_var, (hipStream_t)stream);
}

int icap_bn_finalize_f32(const float* mean, const float* sqdev, const float* gamma, const float* beta, float eps, float momentum,
                         int M, int C, float* rstd, float* s, float* b, float* run_mean, float* run_var, void* stream) {
    return bn_finalize_f32(mean, sqdev, gamma, beta, eps, momentum, M, C, rstd, s, b, run_mean, run_var, (hipStream_t)stream);
}

int icap_scale_bias_act_f32(const float* x, int ldx, const float* s, const float* b, const float* res, int ldr, int act, int M,
                            int C, float* y, int ldy, void* stream) {
    return scale_bias_act_f32(x, ldx, s, b, res, ldr, act, M, C, y, ldy, (hipStream_t)stream);
}

int icap_bn_act_bwd_f32(const float* dy, int ldd, const float* x, int ldx, const float* s, const float* b, const float* mean,
                        const float* rstd, const float* dgamma, const float* dbeta, int act, int M, int C, float* dx, int ldo,
                        void* stream) {
    return bn_act_bwd_f32(dy, ldd, x, ldx, s, b, mean, rstd, dgamma, dbeta, act, M, C, dx, ldo, (hipStream_t)stream);
}

int icap_axpby_f32(const float* x, float alpha, float beta, float* y, long long n, void* stream) {
    return axpby_f32(x, alpha, beta, y, n, (hipStream_t)stream);
}

int icap_dropout_f32(const float* x, float p, unsigned long long seed, unsigned long long offset, long long n, long long group,
                     float* y, float* mask, void* stream) {
    return dropout_f32(x, p, seed, offset, n, group, y, mask, (hipStream_t)stream);
}

int icap_mul_mask_f32(const float* x, const float* mask, long long group, const float* add, long long n, float* y, void* stream) {
    return mul_mask_f32(x, mask, group, add, n, y, (hipStream_t)stream);
}

int icap_ln_train_fwd_f32(const float* x, const float* res, const float* gamma, const float* beta, float eps, float pre_scale,
                          int R, int E, float* y, float* xs, float* mean, float* rstd, void* stream) {
    return ln_train_fwd_f32(x, res, gamma, beta, eps, pre_scale, R, E, y, xs, mean, rstd, (hipStream_t)stream);
}

int icap_ln_bwd_f32(const float* dy, const float* xs, const float* gamma, const float* mean, const float* rstd, float post_scale,
                    int R, int E, float* dx, void* stream) {
    return ln_bwd_f32(dy, xs, gamma, mean, rstd, post_scale, R, E, dx, (hipStream_t)stream);
}

int icap_lstm_bwd_f32(const float* gates, const float* c_in, const float* c_out, const float* dh, const float* dc, int B, int H,
                      float* dgates, float* dc_prev, void* stream) {
    return lstm_bwd_f32(gates, c_in, c_out, dh, dc, B, H, dgates, dc_prev, (hipStream_t)stream);
}

int icap_gru_bwd_f32(const float* gi, const float* gh, const float* h_in, const float* dh, int B, int H, float* dgi, float* dgh,
                     float* dh_direct, void* stream) {
    return gru_bwd_f32(gi, gh, h_in, dh, B, H, dgi, dgh, dh_direct, (hipStream_t)stream);
}

int icap_att_bwd_f32(const float* u, const float* w, const float* a_w, const float* alpha, const float* feat, const float* dctx,
                     int B, int A, int E, int accumulate, float* dscore, float* dw, float* da_part, float* dab_part, float* du,
                     float* dfeat, void* stream) {
    return att_bwd_f32(u, w, a_w, alpha, feat, dctx, B, A, E, accumulate, dscore, dw, da_part, dab_part, du, dfeat, (hipStream_t)stream);
}

int icap_mha_bwd_f32(const float* q, const float* k, const float* v, const float* dO, int R, int Bk, int E, int heads, float* dq,
                     float* dk, float* dv, float* ws, void* stream) {
    return mha_bwd_f32(q, k, v, dO, R, Bk, E, heads, dq, dk, dv, ws, (hipStream_t)stream);
}

int icap_embedding_bwd_f32(const long long* idx, const float* dy, int ldd, int P, int E, int V, float* dtable, void* stream) {
    return embedding_bwd_f32(idx, dy, ldd, P, E, V, dtable, (hipStream_t)stream);
}

int icap_chan_affine_f32(const float* x, const float* g, const float* a, float alpha, int B, int HW, int C, float* y, void* stream) {
    return chan_affine_f32(x, g, a, alpha, B, HW, C, y, (hipStream_t)stream);
}

int icap_image_dot_f32(const float* x, const float* y, int B, int HW, int C, float scale, float* out, float* ws, void* stream) {
    return image_dot_f32(x, y, B, HW, C, scale, out, ws, (hipStream_t)stream);
}
long long icap_image_dot_ws_floats(int B, int HW, int C) { return image_dot_ws_floats(B, HW, C); }

int icap_bn_steps_fwd_f32(const float* x, const float* gamma, const float* beta, float eps, float momentum, int B, int T, int V,
                          float* y, float* mean, float* rstd, float* s, float* b, float* uvar, float* run_mean, float* run_var,
                          void* stream) {
    return bn_steps_fwd_f32(x, gamma, beta, eps, momentum, B, T, V, y, mean, rstd, s, b, uvar, run_mean, run_var, (hipStream_t)stream);
}

int icap_bn_steps_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd, const float* s, int B, int T, int V,
                          float* dgamma_t, float* dbeta_t, float* dx, void* stream) {
    return bn_steps_bwd_f32(dy, x, mean, rstd, s, B, T, V, dgamma_t, dbeta_t, dx, (hipStream_t)stream);
}

int icap_act_pool_f32(const float* x, const float* s, const float* b, int act, int B, int HW, int C, float scale, float* y,
                      float* pooled, float* ws, void* stream) {
    return act_pool_f32(x, s, b, act, B, HW, C, scale, y, pooled, ws, (hipStream_t)stream);
}

long long icap_act_pool_ws_floats(int B, int HW, int C) { return act_pool_ws_floats(B, HW, C); }

int icap_act_gate_f32(const float* x, const float* s, const float* b, int act, const float* g, int B, int HW, int C, float* y,
                      void* stream) {
    return act_gate_f32(x, s, b, act, g, B, HW, C, y, (hipStream_t)stream);
}

int icap_image_dot_act_f32(const float* x, const float* y, const float* s2, const float* b2, int act, int B, int HW, int C, float scale,
                           float* out, float* ws, void* stream) {
    return image_dot_act_f32(x, y, s2, b2, act, B, HW, C, scale, out, ws, (hipStream_t)stream);
}

int icap_bn_act_bwd_se_f32(const float* dy, const float* x, const float* s, const float* b, const float* mean, const float* rstd,
                           const float* gate, const float* dpool, float alpha, int HW, int act, int M, int C, float* dgamma,
                           float* dbeta, float* dx, float* ws, void* stream) {
    return bn_act_bwd_se_f32(dy, x, s, b, mean, rstd, gate, dpool, alpha, HW, act, M, C, dgamma, dbeta, dx, ws, (hipStream_t)stream);
}

int icap_se_train_fwd_f32(const float* pooled, const float* wr, const float* br, const float* we, const float* be, float* r_pre,
                          float* e, float* gate, int B, int C, int S, void* stream) {
    return se_train_fwd_f32(pooled, wr, br, we, be, r_pre, e, gate, B, C, S, (hipStream_t)stream);
}

int icap_se_train_bwd_f32(const float* dgate, const float* e, const float* r_pre, const float* pooled, const float* wr,
                          const float* we, float* de, float* dr_pre, float* dpooled, float* dwe, float* dbe, float* dwr, float* dbr,
                          int B, int C, int S, void* stream) {
    return se_train_bwd_f32(dgate, e, r_pre, pooled, wr, we, de, dr_pre, dpooled, dwe, dbe, dwr, dbr, B, C, S, (hipStream_t)stream);
}

int icap_dwconv_bwd_data_f32(const float* dy, const float* w, int B, int H, int W, int C, int K, int S, int pad_t, int pad_l, int Ho,
                             int Wo, float* dx, void* stream) {
    return dwconv_bwd_data_f32(dy, w, B, H, W, C, K, S, pad_t, pad_l, Ho, Wo, dx, (hipStream_t)stream);
}

int icap_dwconv_bwd_weight_f32(const float* dy, const float* x, int B, int H, int W, int C, int K, int S, int pad_t, int pad_l, int Ho,
                               int Wo, float* dw, float* ws, void* stream) {
    return dwconv_bwd_weight_f32(dy, x, B, H, W, C, K, S, pad_t, pad_l, Ho, Wo, dw, ws, (hipStream_t)stream);
}
long long icap_dwconv_bwd_weight_ws_floats(int B, int Ho, int Wo, int C, int K) { return dwconv_bwd_weight_ws_floats(B, Ho, Wo, C, K); }

int icap_conv2d_bf16_f32(const void* x, const unsigned short* w_bf16, const float* scale, const float* bias, const void* residual,
                         void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho,
                         int Wo, int act, int io_bf16, int cfg, void* stream) {
    return conv2d_bf16_f32(x, w_bf16, scale, bias, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, act, io_bf16,
                           cfg, (hipStream_t)stream);
}

int icap_maxpool2d_bf16(const unsigned short* x, unsigned short* y, int B, int H, int W, int C, int ks, int stride, int pad, int Ho,
                        int Wo, void* stream) {
    return maxpool2d_bf16(x, y, B, H, W, C, ks, stride, pad, Ho, Wo, (hipStream_t)stream);
}

int icap_adaptive_avgpool2d_bf16_f32(const unsigned short* x, float* y, int B, int H, int W, int C, int OH, int OW, void* stream) {
    return adaptive_avgpool2d_bf16_f32(x, y, B, H, W, C, OH, OW, (hipStream_t)stream);
}

long long icap_augment_ws_bytes(int B) { return augment_ws_bytes(B); }
int icap_augment_u8(const unsigned char* src, int B, int H, int W, const int* params, const float* contrast, void* ws,
                    unsigned char* dst, void* stream) {
    return augment_u8(src, B, H, W, params, contrast, static_cast<unsigned long long*>(ws), dst, (hipStream_t)stream);
}

long long icap_resize_normalize_ws_bytes(int B, int H, int Wo) { return resize_normalize_ws_bytes(B, H, Wo); }

int icap_resize_normalize_u8_f32(const unsigned char* src, int B, int H, int W, const int* kx, const int* bx, int ksx, const int* ky,
                                 const int* by, int ksy, int Ho, int Wo, const float* mean_std, void* tmp, float* out, int layout,
                                 void* stream) {
    return resize_normalize_u8_f32(src, B, H, W, kx, bx, ksx, ky, by, ksy, Ho, Wo, mean_std, (unsigned*)tmp, out, layout,
                                   (hipStream_t)stream);
}

int icap_transpose_f32(const float* src, int rows, int cols, int ld_src, float* dst, void* stream) {
    return transpose_f32(src, rows, cols, ld_src, dst, (hipStream_t)stream);
}

int icap_conv_wgrad_gather_f32(const float* dz, int ldz, const float* x, int B, int H, int W, int C, int Cv, int KH, int KW, int S,
                               int pad_t, int pad_l, int Ho, int Wo, int Cout, float* dw, float* ws, void* stream) {
    return conv_wgrad_gather_f32(dz, ldz, x, B, H, W, C, Cv, KH, KW, S, pad_t, pad_l, Ho, Wo, Cout, dw, ws, (hipStream_t)stream);
}

long long icap_conv_wgrad_gather_ws_floats(int B, int Ho, int Wo, int Cout, int N) { return conv_wgrad_gather_ws_floats(B, Ho, Wo, Cout, N); }

int icap_im2col_f32(const float* x, int B, int H, int W, int C, int KH, int KW, int S, int pad_t, int pad_l, int Ho, int Wo, float* col,
                    void* stream) {
    return im2col_f32(x, B, H, W, C, KH, KW, S, pad_t, pad_l, Ho, Wo, col, (hipStream_t)stream);
}

}  // extern "C"
