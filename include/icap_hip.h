/*
 * icap_hip.h - C ABI of libicap_hip.so: the gfx950 (MI355X) kernels behind the
 * captioning hot path of Merumalady/Image_Captioning (CNN encoder forward and the
 * per-step decoder forward of its four models).
 *
 * The reference is pure Python on torch.nn and has no FFI of its own; each entry
 * point below names the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - every tensor argument is a raw DEVICE pointer owned by the caller; the
 *     library never allocates, frees or retains caller memory, workspaces are
 *     passed in;
 *   - fp32 everywhere, token ids are int64 (long long), activations of the CNN are
 *     NHWC, channel counts are multiples of 4 and pointers 16-byte aligned;
 *   - sizes are int (elements), `stream` is a hipStream_t passed as void*;
 *   - calls only enqueue work on `stream` (no synchronisation, no host copies),
 *     so they may be issued from any host thread that has set the device;
 *   - return value 0 = ok, 1 = bad argument, 2 = alignment, 3 = size limit,
 *     1000 + hipError_t = launch failure.  icap_last_error() returns a message for
 *     the calling thread's most recent non-zero return.  Nothing throws.
 */
#ifndef ICAP_HIP_H
#define ICAP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ICAP_ABI_VERSION 1

/* activation codes */
#define ICAP_ACT_NONE 0
#define ICAP_ACT_RELU 1
#define ICAP_ACT_SWISH 2
#define ICAP_ACT_SIGMOID 3
#define ICAP_ACT_TANH 4

int icap_abi_version(void);
/* POSIX cksum (CRC) of the icap_hip.h the library was compiled against, 0 when the build did not record one: a binding that
 * ships with the header compares the two and refuses a library built from another version of this file (a changed
 * signature under an unchanged name would otherwise go unnoticed). */
unsigned icap_header_hash(void);
const char* icap_last_error(void);

/* ------------------------------------------------------------------ encoder */

/* NCHW image batch -> NHWC with channels zero-padded to Cp (Cp % 4 == 0).
 * Replaces the implicit layout of `images` entering the backbone:
 * seq2seqbasic.py:42, transformer_word_level.py:94. */
int icap_nchw_to_nhwc_pad_f32(const float* src, float* dst, int B, int C, int H, int W, int Cp, void* stream);

/* Convolution / Linear as implicit GEMM on the matrix cores:
 *   y = act(conv(x * gate, w) * scale + bias + residual)
 * x [B,H,W,Cin] NHWC, w [Cout,KH,KW,Cin], scale/bias [Cout] (folded BatchNorm or
 * the layer bias; NULL = 1 / 0), residual [B,Ho,Wo,Cout] or NULL, gate [B,Cin]
 * (squeeze-excite scale) or NULL.  Bottom/right padding is implied by Ho, Wo.
 * Replaces torchvision resnet18 / vgg16 convolutions (seq2seqbasic.py:23-34,42),
 * EfficientNet-B0 stem, expand, project and head convolutions
 * (transformer_word_level.py:13,94) and every nn.Linear on the path when called
 * with H = W = KH = KW = 1 (seq2seqbasic.py:43,63,65; transformer_word_level.py:21,
 * 35,38,63,66,69; transformer_word_level_bahdnau.py:98-100,122). */
int icap_conv2d_f32(const float* x, const float* w, const float* scale, const float* bias,
                    const float* residual, const float* gate, float* y, int B, int H, int W, int Cin,
                    int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                    void* stream);

/* Same computation as icap_conv2d_f32, but first times every tile configuration of the kernel on this
 * exact problem and remembers the fastest for later icap_conv2d_f32 calls of the same shape (process-wide).
 * SYNCHRONISES the stream: call once per shape during warm-up.  icap_conv2d_tuned_config returns the remembered
 * configuration index, or -1.  No reference counterpart (the reference leaves kernel choice to cuDNN/MIOpen). */
int icap_conv2d_tune_f32(const float* x, const float* w, const float* scale, const float* bias,
                         const float* residual, const float* gate, float* y, int B, int H, int W, int Cin,
                         int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                         void* stream);
int icap_conv2d_tuned_config(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int Ho, int Wo);

/* Same computation as icap_conv2d_f32 with tile configuration `cfg` in [0, icap_conv2d_num_configs()) instead of the
 * heuristic / tuned choice: the parity tests run every configuration the tuner may select.  Returns 1 (bad argument)
 * when the configuration cannot run the shape (the pipelined K loops need an even number of K steps, some >= 4) and
 * 2 when Cout % 4 != 0 (such outputs always take the one scalar-store form).  No reference counterpart. */
int icap_conv2d_num_configs(void);
/* flag 0: icap_conv2d_f32 ignores the remembered tuning results and uses its shape heuristic (a benchmark re-runs its
 * timed step this way to check the tuned run's output); 1: uses them again (the default); negative: query only.
 * Returns the previous setting. */
int icap_conv2d_use_tuned(int flag);
int icap_conv2d_cfg_f32(const float* x, const float* w, const float* scale, const float* bias,
                        const float* residual, const float* gate, float* y, int B, int H, int W, int Cin,
                        int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho, int Wo, int act,
                        int cfg, void* stream);

/* Depthwise KSxKS convolution (KS = 3 or 5) with fused scale/bias (folded
 * BatchNorm) and activation.  w is [KS*KS][C].  When `partial` is not NULL it
 * receives per-chunk channel sums of the OUTPUT, [B][icap_dwconv_chunks(Ho,Wo,C,KS,stride)][C],
 * for the squeeze-excite pooling.  Replaces MBConv `_depthwise_conv` + `_bn1` +
 * swish of EfficientNet-B0 (transformer_word_level.py:13). */
int icap_dwconv2d_f32(const float* x, const float* w, const float* scale, const float* bias, float* y,
                      float* partial, int B, int H, int W, int C, int KS, int stride, int pad_t, int pad_l,
                      int Ho, int Wo, int act, void* stream);
int icap_dwconv_chunks(int Ho, int Wo, int C, int KS, int stride);

/* Fused MBConv front half: 1x1 expand convolution + folded BatchNorm + swish, then the depthwise KSxKS
 * convolution + folded BatchNorm + swish, with the expanded activations kept on chip.  x [B][H][W][Cin],
 * w_exp [Cexp][Cin], w_dw [KS*KS][Cexp], y [B][Ho][Wo][Cexp]; `partial` as in icap_dwconv2d_f32 with
 * icap_expand_dwconv_chunks(...) chunks per image.  icap_expand_dwconv_chunks returns -1 for layer shapes that
 * have no fused kernel (callers then run icap_conv2d_f32 + icap_dwconv2d_f32).  Replaces MBConv `_expand_conv` +
 * `_bn0` + swish + `_depthwise_conv` + `_bn1` + swish of EfficientNet-B0 (transformer_word_level.py:13). */
int icap_expand_dwconv_f32(const float* x, const float* w_exp, const float* exp_scale, const float* exp_bias,
                           const float* w_dw, const float* dw_scale, const float* dw_bias, float* y, float* partial,
                           int B, int H, int W, int Cin, int Cexp, int KS, int stride, int pad_t, int pad_l, int Ho,
                           int Wo, void* stream);
int icap_expand_dwconv_chunks(int Ho, int Wo, int Cin, int Cexp, int KS, int stride);
/* 1 when icap_expand_dwconv_f32 runs this layer on its whole-image kernel: the stride-1 blocks at 14 x 14 (80 / 112 input
 * channels) and 7 x 7 (192), 3x3 or 5x5 with pad = KS / 2, where a workgroup owns whole images (one 14 x 14 or four 7 x 7)
 * and 64 expanded channels, so no halo is recomputed; one chunk per image (partial [B][1][Cexp] = the channel sums).
 * That kernel takes and gives fp32 tensors only: the bf16 entry point rejects these shapes. */
int icap_expand_dwconv_whole_image(int H, int W, int Cin, int Cexp, int KS, int stride);

/* EfficientNet-B0 stem fused with block 0's depthwise convolution: 3x3 stride-2 convolution of the 4-channel-padded
 * image x [B][img_h][img_w][4] with w_stem [C][3][3][4] (TF-"same" padding stem_pad_t / stem_pad_l at the top / left,
 * output Hs x Ws), folded BatchNorm, swish; then the depthwise 3x3 stride-1 convolution w_dw [9][C] (padding pad_t /
 * pad_l), folded BatchNorm, swish -> y [B][Ho][Wo][C]; `partial` as in icap_dwconv2d_f32 with
 * icap_stem_dwconv_chunks(Ho, Wo, C) chunks per image.  Replaces `_conv_stem` + `_bn0` + swish and block 0's
 * `_depthwise_conv` + `_bn1` + swish (transformer_word_level.py:13). */
int icap_stem_dwconv_f32(const float* x, const float* w_stem, const float* stem_scale, const float* stem_bias,
                         const float* w_dw, const float* dw_scale, const float* dw_bias, float* y, float* partial, int B,
                         int img_h, int img_w, int Hs, int Ws, int C, int stem_pad_t, int stem_pad_l, int pad_t, int pad_l,
                         int Ho, int Wo, void* stream);
/* The same with x the image as the reference hands it over, planar NCHW fp32 [B][3][img_h][img_w] (w_stem stays [C][3][3][4]
 * with a zero fourth channel): saves the pass that would interleave and pad the image first
 * (transformer_word_level.py:99 feeds `self.cnn_model(images)` such a tensor). */
int icap_stem_dwconv_nchw_f32(const float* x, const float* w_stem, const float* stem_scale, const float* stem_bias,
                              const float* w_dw, const float* dw_scale, const float* dw_bias, float* y, float* partial, int B,
                              int img_h, int img_w, int Hs, int Ws, int C, int stem_pad_t, int stem_pad_l, int pad_t, int pad_l,
                              int Ho, int Wo, void* stream);
/* The same layer pair on the vector ALUs alone, for C = 32 stem channels (icap_stem_dwconv_valu_supported): a thread owns one
 * stem pixel and all channels, the stem weights reach the multiply-adds as scalar operands, the depthwise stage slides over
 * the tile in LDS.  x planar NCHW fp32 as in icap_stem_dwconv_nchw_f32; w_stem_k is the stem filter k-major, [27][C] with
 * k = (kh * 3 + kw) * 3 + input channel (the transpose of the reference's [C][3][3][3] weight).  Same result up to the order
 * of the fp32 sums. */
int icap_stem_dwconv_valu_supported(int C);
int icap_stem_dwconv_valu_f32(const float* x, const float* w_stem_k, const float* stem_scale, const float* stem_bias,
                              const float* w_dw, const float* dw_scale, const float* dw_bias, float* y, float* partial, int B,
                              int img_h, int img_w, int Hs, int Ws, int C, int stem_pad_t, int stem_pad_l, int pad_t, int pad_l,
                              int Ho, int Wo, void* stream);
int icap_stem_dwconv_chunks(int Ho, int Wo, int C);

/* icap_expand_dwconv_f32 with its expand GEMM as fp32-ACCURATE products on the bf16 matrix cores (exact three-way bf16
 * splits of both operands in registers, six products per multiply, fp32 accumulation - see icap_linear_f32x3_f32):
 * same arguments, same result to a few fp32 roundings, half the matrix-core cycles of the fp32 instruction. */
int icap_expand_dwconv_f32x3_f32(const float* x, const float* w_exp, const float* exp_scale, const float* exp_bias,
                                 const float* w_dw, const float* dw_scale, const float* dw_bias, float* y, float* partial,
                                 int B, int H, int W, int Cin, int Cexp, int KS, int stride, int pad_t, int pad_l, int Ho,
                                 int Wo, void* stream);

/* The whole-image layer shapes of icap_expand_dwconv_whole_image with split products AND the depthwise stage of one
 * channel chunk issued between the matrix instructions of the next chunk (one wave per SIMD keeps the split input planes
 * of its pixels in registers for the workgroup's life; vector work inside a wave's own MFMA stream runs in the matrix
 * instruction's shadow).  The expand weights come pre-split and packed in fragment order: icap_expand_dwconv_img_x3_pack
 * writes icap_expand_dwconv_img_x3_we_bytes(...) bytes (0 = no such kernel for this shape) from w_exp [Cexp][Cin], once
 * per weight.  Same result as icap_expand_dwconv_f32x3_f32 / icap_expand_dwconv_f32 to a few fp32 roundings; `pooled`
 * [B][Cexp] = per-image channel sums or NULL; pad must be KS / 2.  Same reference lines as icap_expand_dwconv_f32
 * (MBConv `_expand_conv` ... swish of EfficientNet-B0, transformer_word_level.py:13). */
long long icap_expand_dwconv_img_x3_we_bytes(int H, int W, int Cin, int Cexp, int KS);
int icap_expand_dwconv_img_x3_pack(const float* w_exp, void* w_packed, int H, int W, int Cin, int Cexp, int KS, void* stream);
int icap_expand_dwconv_img_x3_f32(const float* x, const void* w_packed, const float* exp_scale, const float* exp_bias,
                                  const float* w_dw, const float* dw_scale, const float* dw_bias, float* y, float* pooled,
                                  int B, int H, int W, int Cin, int Cexp, int KS, int pad_t, int pad_l, void* stream);

/* The same kernel for the block that halves the resolution at 14 x 14 (112 -> 672 channels, 5 x 5 depthwise, stride 2,
 * TF-'same' padding: pad_t = pad_l = 1): x [B][14][14][Cin] -> y [B][7][7][Cexp], pooled [B][Cexp] or NULL; w_packed is
 * what icap_expand_dwconv_img_x3_pack writes for (14, 14, Cin, Cexp, KS).  Replaces the pair icap_linear_f32x3_f32
 * (expand + BN + swish) + icap_dwconv2d_f32: the expanded 14 x 14 tensor never reaches HBM.  icap_..._s2_supported
 * returns 1 for the layer shapes that have a kernel.  Reference: MBConv block 11 of EfficientNet-B0
 * (`_expand_conv` ... `_depthwise_conv` stride 2, transformer_word_level.py:13). */
int icap_expand_dwconv_img_x3_s2_supported(int H, int W, int Cin, int Cexp, int KS, int pad_t, int pad_l);
int icap_expand_dwconv_img_x3_s2_f32(const float* x, const void* w_packed, const float* exp_scale, const float* exp_bias,
                                     const float* w_dw, const float* dw_scale, const float* dw_bias, float* y, float* pooled,
                                     int B, int H, int W, int Cin, int Cexp, int KS, int pad_t, int pad_l, void* stream);

/* The two fused kernels with bf16 products in their first (expand / stem) convolution - operands rounded to bf16 in
 * registers, fp32 accumulation on v_mfma_f32_32x32x16_bf16; the depthwise stage computes in fp32.  y is fp32
 * (y_bf16 = 0) or a bf16 tensor of the same shape (y_bf16 = 1, round to nearest even; the squeeze-excite partial sums
 * are taken before the rounding).  Otherwise the arguments and layouts of the fp32 forms above. */
int icap_expand_dwconv_bf16_f32(const float* x, const float* w_exp, const float* exp_scale, const float* exp_bias,
                                const float* w_dw, const float* dw_scale, const float* dw_bias, void* y, int y_bf16,
                                float* partial, int B, int H, int W, int Cin, int Cexp, int KS, int stride, int pad_t,
                                int pad_l, int Ho, int Wo, void* stream);
int icap_stem_dwconv_bf16_f32(const float* x, const float* w_stem, const float* stem_scale, const float* stem_bias,
                              const float* w_dw, const float* dw_scale, const float* dw_bias, void* y, int y_bf16,
                              float* partial, int B, int img_h, int img_w, int Hs, int Ws, int C, int stem_pad_t,
                              int stem_pad_l, int pad_t, int pad_l, int Ho, int Wo, void* stream);

/* Squeeze-excite gate: gate[b][c] = sigmoid(We . swish(Wr . mean_hw(x) + br) + be),
 * from the chunk sums written by icap_dwconv2d_f32.  wr [S][C]; we [S][C] (the expand weight TRANSPOSED).
 * Replaces MBConv `_se_reduce` / `_se_expand` (transformer_word_level.py:13). */
int icap_se_gate_f32(const float* partial, int nchunks, float inv_hw, const float* wr, const float* br,
                     const float* we, const float* be, float* gate, int B, int C, int S, void* stream);

/* Max pooling, square window, symmetric padding (resnet18 maxpool 3/2/1, vgg16
 * 2/2/0: seq2seqbasic.py:24,33). */
int icap_maxpool2d_f32(const float* x, float* y, int B, int H, int W, int C, int KS, int stride, int pad,
                       int Ho, int Wo, void* stream);

/* Adaptive average pooling to OH x OW (1x1 = global pool of resnet18 /
 * EfficientNet-B0, 7x7 for vgg16: seq2seqbasic.py:24,34). */
int icap_adaptive_avgpool2d_f32(const float* x, float* y, int B, int H, int W, int C, int OH, int OW, void* stream);

/* ------------------------------------------------------------------ decoder rows */

/* out[r] = table[idx[r]] (+ add[r]).  *err_flag is set to 1 on an out-of-range id.
 * nn.Embedding: seq2seqbasic.py:61,88; transformer_word_level.py:105;
 * transformer_word_level_bahdnau.py:133. */
int icap_embedding_f32(const float* table, const long long* idx, const float* add, float* out, int rows,
                       int E, int V, int* err_flag, void* stream);

/* y[r] = LayerNorm(x[r] * pre_scale + res[r % res_rows]) * gamma + beta.
 * transformer_word_level.py:46,48,75,77; transformer_word_level_bahdnau.py:77,88. */
int icap_layernorm_f32(const float* x, const float* res, int res_rows, const float* gamma, const float* beta,
                       float* y, int rows, int E, float eps, float pre_scale, void* stream);
/* The same LayerNorm, its result also (y3) or only (y NULL) as the exact three-way bf16 split icap_split3_f32 would make of
 * it, planes [3][rows*E]: the pre-split activation of icap_linear_f32x3_argmax_f32 without a pass of its own (the greedy
 * loop's LayerNorm in front of the vocabulary projection; same reference lines). */
int icap_layernorm_split3_f32(const float* x, const float* res, int res_rows, const float* gamma, const float* beta,
                              float* y, unsigned short* y3, int rows, int E, float eps, float pre_scale, void* stream);

/* idx[r*idx_stride] = argmax(logits[r][0..V)) (first maximum); top2gap[r] (optional)
 * = best - runner-up.  seq2seqbasic.py:81 and the greedy form of TransformerTest.py:147-157. */
int icap_argmax_rows_f32(const float* logits, int ld, long long* idx, int idx_stride, int rows, int V,
                         float* top2gap, void* stream);

/* bf16-compute form of a Linear layer / 1x1 convolution (fp32 tensors in memory, bf16 weights, fp32 accumulation on
 * v_mfma_f32_32x32x16_bf16):  y[M][N] = act((x[M][K] (* gate[m / hw][k]) @ w^T) * scale[n] + bias[n] (+ residual[m][n])),
 * w_bf16 [N][K] as produced by icap_to_bf16_f32 (round to nearest even); x is fp32, or bf16 when x_bf16 = 1 (the bf16
 * result of a fused kernel).  K % 8 == 0, N % 4 == 0.  cfg -1 picks the
 * tile shape (0: 128x128, 1: 128x64, 2: 64x64, 3: 128x256).  Same layers as icap_conv2d_f32 with KH = KW = 1 - MBConv `_expand_conv`,
 * `_project_conv`, `_conv_head` (transformer_word_level.py:13) and every nn.Linear of the decoders - at the precision
 * north-star configuration names for the accelerated path. */
int icap_linear_bf16_f32(const void* x, int x_bf16, const unsigned short* w_bf16, const float* scale, const float* bias,
                         const float* residual, const float* gate, int hw, float* y, int M, int K, int N, int act, int cfg,
                         void* stream);
int icap_to_bf16_f32(const float* src, unsigned short* dst, long long n, void* stream);

/* The thin 1x1 "projection" convolutions of the high-resolution MBConv blocks (transformer_word_level.py:13: EfficientNet-B0's
 * _project_conv + _bn2 of blocks 0-2: 32 -> 16, 96 / 144 -> 24 channels over 3.2 - 12.8 M pixels at B = 1024) as a streaming
 * kernel: y[M][N] = ((x[M][K] * gate[m / hw][k]) @ w[N][K]^T) * scale[n] + bias[n] (+ residual[m][n]) with the whole weight
 * matrix resident in LDS, rows loaded straight into registers in the fp32 MFMA's operand layout, no barrier in the loop.
 * Exact fp32 products (v_mfma_f32_32x32x2_f32).  icap_linear_thin_supported(K, N) = 1 for K in {32, 96, 144} with N <= 32,
 * N % 4 == 0; gate (optional) needs hw >= 32; M < 2^31.  Same results as icap_conv2d_f32 on the same
 * layer up to the order of the fp32 sums. */
int icap_linear_thin_supported(int K, int N);
int icap_linear_thin_f32(const float* x, const float* w, const float* scale, const float* bias, const float* residual,
                         const float* gate, int hw, float* y, long long M, int K, int N, void* stream);

/* fp32-ACCURATE linear layers on the bf16 matrix cores (the fp32 MFMA retires 1/16 of the bf16 MFMA's flops per cycle).
 * Every fp32 operand is the exact sum of three bf16 values (hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid));
 * the six products hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi are exact in the MFMA's fp32 accumulation and the three
 * dropped ones are each <= 2^-24 of the product, so the result differs from an fp32 dot product by a few units of fp32
 * rounding.  icap_split3_f32 prepares the weights once: src fp32 [n] -> dst bf16 [3][n] (n even).  The activations x stay
 * fp32 in memory and are split while they are staged (x_split = 0), or arrive as the three planes [3][M][K] icap_split3_f32
 * makes of them (x_split = 1, no gate): worth it when the result has many column tiles, each of which would repeat the
 * split arithmetic (the decoder's vocabulary projection: 53).  Same epilogue as icap_conv2d_f32 / icap_linear_argmax_f32
 * (scale, bias, residual, gate [M/hw][K], activation; arg-max partials per 128-column tile:
 * icap_linear_bf16_argmax_parts(N) per row).  K % 8 == 0, N % 4 == 0.  cfg: -1 = by shape, 0 = 128x128, 1 = 64x128,
 * 2 = 64x64 tiles, 3 = the kernel for tall activations (the gated 14 x 14 / 7 x 7 projection convolutions: a wave keeps
 * its 32 rows in registers for the whole contraction, split there, and only the weights pass through LDS; fp32 x, no
 * activation function, shapes for which icap_linear_f32x3_tall_supported(K, N, hw, gated) = 1: K a multiple of 48, gated
 * images of at least 43 pixels).  Replaces the same nn.Linear / 1x1 convolution sites as icap_conv2d_f32. */
int icap_split3_f32(const float* src, unsigned short* dst, long long n, void* stream);
int icap_linear_f32x3_tall_supported(int K, int N, int hw, int gated);
/* icap_linear_f32x3_f32 (fp32 x) with the contraction cut into ksplit (2 .. 16) slices of whole 32-wide steps, for layers with
 * few output tiles and a long contraction (the word decoder's 2048 -> 512 Linear at 1024 rows, transformer_word_level.py:77:
 * 128 tiles on 256 CUs): ksplit times as many workgroups write partial products to ws (icap_linear_f32x3_splitk_ws_floats(M, N,
 * ksplit) floats, caller-owned, overwritten), a second launch adds the slices in a fixed order and applies scale, bias,
 * residual and activation.  cfg 0 .. 2 as in icap_linear_f32x3_f32 (no -1). */
long long icap_linear_f32x3_splitk_ws_floats(int M, int N, int ksplit);
int icap_linear_f32x3_splitk_f32(const float* x, const unsigned short* w3, const float* scale, const float* bias, const float* residual,
                                 const float* gate, int hw, float* y, float* ws, long long ws_floats, int M, int K, int N, int act,
                                 int cfg, int ksplit, void* stream);
int icap_linear_f32x3_f32(const void* x, int x_split, const unsigned short* w3, const float* scale, const float* bias,
                          const float* residual, const float* gate, int hw, float* y, int M, int K, int N, int act, int cfg,
                          void* stream);
int icap_linear_f32x3_argmax_f32(const void* x, int x_split, const unsigned short* w3, const float* scale, const float* bias,
                                 float* y, float* part_val, int* part_idx, int M, int K, int N, void* stream);

/* The same fp32-accurate split-product Linear with BOTH operands pre-split (x3 [3][M][K], w3 [3][N][K] from icap_split3_f32)
 * and streamed global -> LDS without passing through registers (global_load_lds, a three-stage ring, one wait and one
 * barrier per K slice of 32): the GEMMs of the batched greedy decoder (transformer_word_level.py:76-78: fc1, fc2, output_fc
 * at M = images), whose activations are tiny against the weights.  y fp32 [M][N] and / or y3 - the result already as its
 * three bf16 planes [3][M][N], i.e. the next GEMM's x3.  K % 32 == 0, N % 4 == 0.  cfg: -1 = by shape, 0 = 128x128 tiles,
 * 1 = 64x64.  The arg-max form emits partials per 128-column tile (icap_linear_bf16_argmax_parts(N) per row). */
int icap_linear_x3dl_f32(const unsigned short* x3, const unsigned short* w3, const float* scale, const float* bias, const float* residual,
                         float* y, unsigned short* y3, int M, int K, int N, int act, int cfg, void* stream);
int icap_linear_x3dl_argmax_f32(const unsigned short* x3, const unsigned short* w3, const float* scale, const float* bias, float* y,
                                float* part_val, int* part_idx, int M, int K, int N, void* stream);

/* bf16-product form of icap_conv2d_f32 for the dense KH x KW convolutions of the ResNet-18 / VGG-16 encoders
 * (seq2seqbasic.py:19-35: torchvision conv3x3 + BN + ReLU, residual add): x [B][H][W][Cin] NHWC with Cin % 32 == 0,
 * w_bf16 [Cout][KH][KW][Cin] from icap_to_bf16_f32 - or Cin == 4 (the channel-padded RGB image of the first layer) with
 * every weight row zero-padded from KH*KW*4 to the next multiple of 32 values - fp32 accumulation,
 * y = act(conv * scale + bias (+ residual)) [B][Ho][Wo][Cout]; taps outside the image are zeros (pad_t / pad_l; bottom /
 * right padding follows from Ho / Wo).  Tensors are fp32 unless io_bf16 says otherwise: bit 0 = x is bf16 (Cin % 32 == 0
 * only), bit 1 = y is stored as bf16 (round to nearest even), bit 2 = residual is bf16 - activations kept in bf16 from
 * one layer to the next.  cfg: -1 = by shape, 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 128x256, 4 = 256x128 tiles of the
 * four-wave kernel; 5 = the eight-wave kernel on 256-row tiles whose operands go global -> LDS directly (bf16 x and y, bf16
 * or no residual, Cin % 64 == 0, Cout % 8 == 0 - anything else is an argument error; cfg -1 picks it for such layers from
 * 192 tiles and four 64-deep K tiles up). */
int icap_conv2d_bf16_f32(const void* x, const unsigned short* w_bf16, const float* scale, const float* bias, const void* residual,
                         void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_t, int pad_l, int Ho,
                         int Wo, int act, int io_bf16, int cfg, void* stream);

/* icap_maxpool2d_f32 / icap_adaptive_avgpool2d_f32 (torchvision MaxPool2d, AdaptiveAvgPool2d: seq2seqbasic.py:21-35) for
 * activations stored in bf16: max pooling bf16 -> bf16 (exact; C % 8 == 0), average pooling bf16 -> fp32 (C % 4 == 0). */
int icap_maxpool2d_bf16(const unsigned short* x, unsigned short* y, int B, int H, int W, int C, int ks, int stride, int pad, int Ho,
                        int Wo, void* stream);
int icap_adaptive_avgpool2d_bf16_f32(const unsigned short* x, float* y, int B, int H, int W, int C, int OH, int OW, void* stream);

/* icap_linear_bf16_f32 (no residual / gate / activation) whose epilogue also writes arg-max partials: per row and
 * 128-column tile the maximum and its column, part_val / part_idx [M][icap_linear_bf16_argmax_parts(N)], consumed by
 * icap_next_token_f32; y may be NULL.  The bf16-mode form of icap_linear_argmax_f32 (greedy TransformerTest.py:147-157). */
int icap_linear_bf16_argmax_f32(const float* x, const unsigned short* w_bf16, const float* scale, const float* bias, float* y,
                                float* part_val, int* part_idx, int M, int K, int N, void* stream);
int icap_linear_bf16_argmax_parts(int N);

/* Top-k sampling head: per row softmax(logits[r][0..V)), its k (<= 16) most probable tokens - probs [rows][k] and
 * idx [rows][k], most probable first - and ids[r*ids_stride] = one of them drawn with probability proportional to its
 * softmax probability from the counter-based generator (row r of a call uses uniform(seed, offset + r), so a caller
 * advances offset by `rows` per call).  probs / idx / ids are each optional.  The reference's generation mode:
 * F.softmax -> torch.topk(probs, 10) -> torch.multinomial (TransformerTest.py:147-157). */
int icap_topk_sample_f32(const float* logits, int ld, int rows, int V, int k, unsigned long long seed,
                         unsigned long long offset, float* probs, int* idx, long long* ids, int ids_stride, void* stream);

/* Scaled-dot-product attention after the input projections, sequence-first layout:
 * q [Lq][N][E], k/v [Lk][Nk][E] with Nk = N or 1.  nn.MultiheadAttention core:
 * transformer_word_level.py:45,74; transformer_word_level_bahdnau.py:74. */
int icap_mha_core_f32(const float* q, const float* k, const float* v, float* out, int Lq, int Lk, int N,
                      int Nk, int E, int heads, void* stream);

/* Additive attention over the images of a batch: u = U(features) [B][A],
 * w = W(hidden) [B][A]; alpha [B][B] (optional), context [B][E].
 * transformer_word_level_bahdnau.py:102-110. */
int icap_bahdanau_attention_f32(const float* u, const float* w, const float* a_w, float a_b, const float* feat,
                                float* alpha, float* context, int B, int A, int E, void* stream);

/* Gate non-linearities after a batched gate GEMM.  LSTM gates [B][4H] (i,f,g,o,
 * biases included); GRU gi/gh [B][3H] (r,z,n).  torch.nn.LSTM/GRU/LSTMCell:
 * seq2seqbasic.py:54,56,64,79; transformer_word_level_bahdnau.py:119,136. */
int icap_lstm_pointwise_f32(const float* gates, const float* c_in, float* h_out, float* c_out, int B, int H, void* stream);
int icap_gru_pointwise_f32(const float* gi, const float* gh, const float* h_in, float* h_out, int B, int H, void* stream);

/* ------------------------------------------------------------------ decoder, batched (GEMM path) greedy step */

/* Vocabulary projection with fused arg-max partials: y[M][N] = x[M][K] . w[N][K]^T * scale + bias (y may be
 * NULL), plus per row and 64-column tile the maximum and its column, part_val / part_idx
 * [M][icap_linear_argmax_parts(N)].  N % 4 == 0.  nn.Linear + argmax of one greedy step for a batch of
 * independent captions: transformer_word_level.py:78 + the argmax form of TransformerTest.py:147-157;
 * transformer_word_level_bahdnau.py:137 (+ folded BatchNorm1d). */
int icap_linear_argmax_f32(const float* x, const float* w, const float* scale, const float* bias, float* y,
                           float* part_val, int* part_idx, int M, int K, int N, void* stream);
int icap_linear_argmax_parts(int N);

/* Finish the arg-max (ids[r*ids_stride] = token) and build the next step's input row
 * out[r] = LayerNorm(table[token] + add[r]) (add optional; LayerNorm only when gamma != NULL; table NULL = ids only).
 * transformer_word_level.py:105,75 (caption embedding, x + cross-attention, layer_norm2);
 * transformer_word_level_bahdnau.py:133 (embedding). */
int icap_next_token_f32(const float* part_val, const int* part_idx, int nparts, long long* ids, int ids_stride,
                        const float* table, const float* add, const float* gamma, const float* beta, float eps,
                        float* out, int ldo, int rows, int E, void* stream);

/* ------------------------------------------------------------------ decoder, small-batch weight streaming */

/* y[b][n] = act((W[n,:].x[b,:]) * scale[n] + bias[n]) for NB <= 8 rows of x.
 * With part_val/part_idx not NULL also writes per-workgroup arg-max partials
 * [NB][icap_gemv_blocks(N)] (y may then be NULL).  Linear layers of one decode
 * step: seq2seqbasic.py:80,89; transformer_word_level.py:76,78. */
int icap_gemv_f32(const float* W, const float* scale, const float* bias, const float* x, int ldx, float* y,
                  int ldy, int NB, int N, int K, int act, float* part_val, int* part_idx, void* stream);
int icap_gemv_blocks(int N);

/* Reduce arg-max partials to ids[b*ids_stride]; optionally gather the next input
 * row next_x[b] = table[id] (+ add[b]).  seq2seqbasic.py:81,88-89. */
int icap_argmax_finalize_f32(const float* part_val, const int* part_idx, int nparts, long long* ids,
                             int ids_stride, const float* table, const float* add, float* next_x, int ldx,
                             int E, int NB, void* stream);

/* One LSTM / GRU cell step for NB <= 8 sequences, weights in torch layout
 * ([4H or 3H][I], [4H or 3H][H], two bias vectors).  h_out must not alias h_in.
 * seq2seqbasic.py:79; transformer_word_level_bahdnau.py:136. */
int icap_lstm_cell_f32(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                       const float* x, int ldx, const float* h_in, const float* c_in, float* h_out,
                       float* c_out, int NB, int I, int H, void* stream);
int icap_gru_cell_f32(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                      const float* x, int ldx, const float* h_in, float* h_out, int NB, int I, int H,
                      void* stream);

/* ------------------------------------------------------------------ greedy drivers
 * Each enqueues `steps` decode steps for NB <= 8 captions in lock-step without any
 * host synchronisation; ids [NB][steps] receives every step's token (the caller cuts
 * at the end token).  logits_out (optional) [steps][NB][V] receives every step's
 * logits.  All pointers in the descriptor structs are device pointers. */

typedef struct {
    int layers, H, I, V, is_lstm;            /* I = input width of layer 0 */
    const float* w_ih[4]; const float* w_hh[4]; const float* b_ih[4]; const float* b_hh[4];
    const float* fc_w; const float* fc_b;    /* [V][H], [V] */
    const float* in_table;                   /* [V][I]: feature_fc(embed) rows */
} icap_rnn_weights;

/* DecoderRNN.sample (seq2seqbasic.py:68-91).  x0 [NB][I] is feature_fc(features).
 * ws: float workspace, icap_greedy_rnn_ws_floats(NB, layers, H, I, V) floats. */
int icap_greedy_rnn_f32(const icap_rnn_weights* w, const float* x0, long long* ids, float* logits_out, float* ws,
                        int NB, int steps, void* stream);
long long icap_greedy_rnn_ws_floats(int NB, int layers, int H, int I, int V);

/* icap_greedy_rnn_f32 for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): no `ws` argument - the workspace (recurrent state h/c, arg-max partials) is a
 * fixed ICAP_GREEDY_RNN_STATIC_WS_FLOATS-float array inside the library's device image, one copy per GPU held until the
 * process ends; the call allocates and frees nothing.  Returns 3 (size limit) when
 * icap_greedy_rnn_ws_floats(NB, layers, H, I, V) exceeds that size.  All calls on one GPU share the array: calls that
 * overlap on one GPU (two streams, two threads) race - serialise them.  Results are identical to icap_greedy_rnn_f32.
 * No counterpart in the reference. */
#define ICAP_GREEDY_RNN_STATIC_WS_FLOATS (1 << 18)
int icap_greedy_rnn_static_ws_floats(void);
int icap_greedy_rnn_static_f32(const icap_rnn_weights* w, const float* x0, long long* ids, float* logits_out, int NB, int steps,
                               void* stream);

typedef struct {
    int E, FF, V;
    const float* emb;                        /* caption_embedding [V][E] */
    const float* ln2_g; const float* ln2_b; const float* ln3_g; const float* ln3_b;
    const float* fc1_w; const float* fc1_b;  /* [FF][E] */
    const float* fc2_w; const float* fc2_b;  /* [E][FF] */
    const float* out_w; const float* out_b;  /* [V][E] */
} icap_tfm_weights;

/* Greedy form of generate_caption for the word / char transformer
 * (TransformerTest.py:139-164 over transformer_word_level.py:72-79): position t's
 * logits depend only on token t and the image, so one step is
 * LN2(emb[tok] + ca) -> FFN -> LN3 -> output_fc -> argmax.  ca [NB][E] is the
 * per-image cross-attention output.  start_ids [NB]. */
int icap_greedy_tfm_f32(const icap_tfm_weights* w, const float* ca, const long long* start_ids, long long* ids,
                        float* logits_out, float* ws, int NB, int steps, void* stream);
long long icap_greedy_tfm_ws_floats(int NB, int E, int FF, int V);

/* Greedy decoding as a walk over a per-image next-token table.  The reference's transformer decoders use neither
 * self-attention nor positional encoding (transformer_word_level.py:72-79, transformercharmiguel.py:73-80: only
 * cross_attention, the FFN and output_fc are called), so the logits after a token depend on that token and the image
 * only; for a small vocabulary (the 81-symbol char model) the whole table next[v][b] = argmax(logits(token v, image b))
 * comes out of ONE batched pass of the decoder over V rows per image, and the T-step caption of generate_caption
 * (TransformerTest.py:139-164, argmax form) is T look-ups.  next int64 [V][B], start int64 [B], ids int64 [B][steps];
 * logits_out (optional) [steps][B][V] is gathered from logits_tab [V*B][ldl] (row v*B + b).  V + steps <= 15360. */
int icap_table_walk(const long long* next, const long long* start, long long* ids, const float* logits_tab, int ldl,
                    float* logits_out, int V, int B, int steps, void* stream);

/* The same greedy decode as icap_greedy_tfm_f32 as ONE persistent launch for the whole caption: the step's weight rows are
 * split over a grid of workgroups and stay in their LDS for all `steps`; only the activation vectors cross between
 * workgroups (64-bit {value, tag} words polled with device-scope atomic loads - no kernel boundary and no grid barrier
 * inside a step).  For the decoders whose per-step weights fit the chip's LDS: the char model (transformercharmiguel.py:
 * 55-80, 2.3 MB over 32 workgroups) and the word model (transformer_word_level.py:54-79, 22 MB over 256).
 * icap_greedy_tfm_persist_max_nb: captions per call the shape allows (0 = shape not supported: use icap_greedy_tfm_f32);
 * ws: icap_greedy_tfm_persist_ws_bytes(NB, E, FF, V) bytes, 8-byte aligned (cleared by the call).  If a workgroup
 * cannot make progress (the grid is not co-resident) the spin limit ends the launch and every id is written as -1. */
int icap_greedy_tfm_persist_max_nb(int E, int FF, int V);
long long icap_greedy_tfm_persist_ws_bytes(int NB, int E, int FF, int V);
int icap_greedy_tfm_persist_f32(const icap_tfm_weights* w, const float* ca, const long long* start_ids, long long* ids,
                                float* logits_out, void* ws, int NB, int steps, void* stream);

/* The Bahdanau LSTMCell decoder's greedy caption (icap_greedy_bah_fused_f32's step: table row for the token-fed input side,
 * W_hh . h computed a step ahead) as ONE persistent launch: the vocabulary projection and W_hh stay in the LDS of one
 * workgroup per CU for all `steps`, and a step is a single exchange of {value, tag} words between the workgroups (partial
 * arg-max words polled, the 4H recurrent values read once and verified by tag; double-buffered by step parity).  Operands as
 * icap_greedy_bah_fused_f32 (g_tab [V][4H], g_ctx [NB][4H], w_hh [4H][H], fc_w [V][H], optional fc_scale / fc_bias).
 * icap_greedy_bah_persist_max_nb: captions per call (at most 2; 0 = the shape does not fit the chip's LDS);
 * ws: icap_greedy_bah_persist_ws_bytes bytes, 8-byte aligned, cleared by the call.  The grid must be co-resident (a plain
 * launch on an otherwise idle device); if a workgroup cannot make progress a bounded spin ends the launch and every id is -1.
 * Replaces the generate_caption loop of TransformerTest_bahdnau.py:153-191 (argmax form) for one image. */
int icap_greedy_bah_persist_max_nb(int H, int V);
long long icap_greedy_bah_persist_ws_bytes(int NB, int H, int V);
int icap_greedy_bah_persist_f32(const float* g_tab, const float* g_ctx, const float* w_hh, const float* fc_w, const float* fc_scale,
                                const float* fc_bias, const long long* start_ids, long long* ids, float* logits_out, void* ws,
                                int NB, int H, int V, int steps, void* stream);

typedef struct {
    int E, H, V;                             /* embed dim, decoder dim, vocab */
    const float* emb;                        /* [V][E] */
    const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh; /* LSTMCell(E+enc -> H) */
    const float* fc_w;                       /* [V][H] */
    const float* fc_scale; const float* fc_bias; /* Linear bias and eval BatchNorm1d(V) folded */
} icap_bah_weights;

/* Greedy form of generate_caption for the Bahdanau model at one image per caption
 * (TransformerTest_bahdnau.py:166-191 over transformer_word_level_bahdnau.py:127-141):
 * with a single image the attention weight is 1 and the context is the encoder row.
 * context [NB][ENC]. */
int icap_greedy_bah_f32(const icap_bah_weights* w, const float* context, int ENC, const long long* start_ids,
                        long long* ids, float* logits_out, float* ws, int NB, int steps, void* stream);
long long icap_greedy_bah_ws_floats(int NB, int E, int ENC, int H, int V);

/* The three step-per-launch drivers above with bf16 WEIGHT STREAMING: at 1..8 captions a step's time is the time to read
 * every weight once (seq2seqbasic.py:79-80: the LSTM / GRU matrices and Linear(H, V); transformer_word_level.py:76-78: fc1,
 * fc2, output_fc; transformer_word_level_bahdnau.py:136-139: LSTMCell and fc), so these variants read half the bytes.
 * Same descriptor structs and workspaces, but the MATRIX members - w_ih[], w_hh[], fc_w (rnn); fc1_w, fc2_w, out_w (tfm);
 * w_ih, w_hh, fc_w (bah) - point at bf16 copies (icap_to_bf16_f32 of the fp32 matrices: round to nearest even, same
 * row-major layout; row lengths must be multiples of 8).  Embedding tables, in_table, biases, LayerNorm / BatchNorm
 * parameters, inputs, state, accumulation and outputs stay fp32: the only difference from the fp32 drivers is one
 * rounding of each weight to 8 significand bits.  Error 2 (alignment) when a row length is not a multiple of 8. */
int icap_greedy_rnn_w16_f32(const icap_rnn_weights* w, const float* x0, long long* ids, float* logits_out, float* ws,
                            int NB, int steps, void* stream);
int icap_greedy_tfm_w16_f32(const icap_tfm_weights* w, const float* ca, const long long* start_ids, long long* ids,
                            float* logits_out, float* ws, int NB, int steps, void* stream);
int icap_greedy_bah_w16_f32(const icap_bah_weights* w, const float* context, int ENC, const long long* start_ids,
                            long long* ids, float* logits_out, float* ws, int NB, int steps, void* stream);

/* Single-caption latency forms of the recurrent greedy drivers: fewer DEPENDENT launches per step.  The input side of a
 * cell as a function of the token is one row of a [V][G*H] table made once per weight version (g_tab = W_ih . in(token) +
 * biases, G = 4 gates for an LSTM, 3 for a GRU); the recurrent side W_hh . h(t) is computed by the launch that projects
 * h(t) on the vocabulary, as extra row segments, a step ahead of its use; the cell update itself (a gather and a pointwise
 * step) is redone in the prologue of every workgroup of the next weight-streaming launch.  Results equal the per-cell
 * drivers' up to the order of the three additions that form a gate pre-activation.  NB <= 8, meant for NB <= 2 (the
 * prologue is redone by hundreds of workgroups).
 *
 * icap_greedy_bah_fused_f32: ONE launch per step for the Bahdanau LSTMCell decoder (icap_greedy_bah_f32: two).
 *   g_tab [V][4H] = emb . W_ih[:, :E]^T;  g_ctx [NB][4H] = context . W_ih[:, E:]^T + b_ih + b_hh (once per caption);
 *   w_hh [4H][H]; fc_w [V][H] with fc_scale / fc_bias as in icap_bah_weights.  Same reference lines as icap_greedy_bah_f32
 *   (TransformerTest_bahdnau.py:166-191 over transformer_word_level_bahdnau.py:127-141).
 * icap_greedy_rnn2_fused_f32: TWO launches per step for the two-layer LSTM / GRU of DecoderRNN.sample (seq2seqbasic.py:68-91;
 *   icap_greedy_rnn_f32: three).  g_tab0 [V][G*H] = in_table . W_ih0^T + b_ih0 (LSTM: + b_hh0); g_first [NB][G*H] the same
 *   for the image feature, step 0's input; hh0_bias / hh1_bias = b_hh of the layer for a GRU (its n gate keeps the two
 *   halves apart), NULL for an LSTM; ih1_bias = b_ih1 (LSTM: + b_hh1). */
/* icap_greedy_rnn2_persist_f32: the two-layer LSTM / GRU of icap_greedy_rnn2_fused_f32 (same operands) as ONE persistent launch
 * for a whole caption: W_ih1, W_hh0, W_hh1 and the vocabulary projection stay in the LDS of a grid of workgroups (one per
 * CU), a step is two exchanges through 64-bit {value, tag} words in `ws` (icap_greedy_rnn2_persist_ws_bytes bytes, 8-byte
 * aligned; cleared by the call).  NB <= icap_greedy_rnn2_persist_max_nb(is_lstm, H, V) (0: the shape does not fit; at most 2).
 * All workgroups must be resident at once: a launch that cannot make progress gives up after a bounded spin and writes -1
 * into every id.  ids [NB][steps]; logits_out [steps][NB][V] or NULL.  Same reference lines as icap_greedy_rnn_f32
 * (seq2seqbasic.py:68-91). */
int icap_greedy_rnn2_persist_max_nb(int is_lstm, int H, int V);
long long icap_greedy_rnn2_persist_ws_bytes(int NB, int is_lstm, int H, int V);
int icap_greedy_rnn2_persist_f32(int is_lstm, const float* g_tab0, const float* g_first, const float* w_hh0,
                                 const float* hh0_bias, const float* w_ih1, const float* ih1_bias, const float* w_hh1,
                                 const float* hh1_bias, const float* fc_w, const float* fc_b, long long* ids,
                                 float* logits_out, void* ws, int NB, int H, int V, int steps, void* stream);
long long icap_greedy_bah_fused_ws_floats(int NB, int H, int V);
int icap_greedy_bah_fused_f32(const float* g_tab, const float* g_ctx, const float* w_hh, const float* fc_w,
                              const float* fc_scale, const float* fc_bias, const long long* start_ids, long long* ids,
                              float* logits_out, float* ws, int NB, int H, int V, int steps, void* stream);
long long icap_greedy_rnn2_fused_ws_floats(int NB, int is_lstm, int H, int V);
int icap_greedy_rnn2_fused_f32(int is_lstm, const float* g_tab0, const float* g_first, const float* w_hh0,
                               const float* hh0_bias, const float* w_ih1, const float* ih1_bias, const float* w_hh1,
                               const float* hh1_bias, const float* fc_w, const float* fc_b, long long* ids,
                               float* logits_out, float* ws, int NB, int H, int V, int steps, void* stream);

/* ------------------------------------------------------------------ training step: losses and optimizer */

/* Masked cross entropy over R rows of V logits.  Row r of the logits lives at
 * ((r / group_rows) * group_stride + r % group_rows) * ld floats, which addresses
 * predictions[:, :-1] of a [B][T][V] tensor in place (group_rows = T-1,
 * group_stride = T); a plain [R][V] matrix is group_rows = group_stride = R.
 * mean_over_valid = 0: sum over non-ignored rows / R, pads stay in the denominator
 *   (compute_loss, data.py:244-252, datachar.py:234-242);
 * mean_over_valid = 1: sum / number of non-ignored rows
 *   (nn.CrossEntropyLoss(ignore_index=PAD), seq2seqbasic.py:204,219).
 * row_loss, row_lse [R] are kept for the backward pass; out[0] = loss, out[1] = the
 * per-row scale that was applied. */
int icap_ce_loss_fwd_f32(const float* logits, int ld, int group_rows, int group_stride, const long long* targets,
                         int R, int V, long long ignore_index, int mean_over_valid, float* row_loss, float* row_lse,
                         float* out, void* stream);

/* dlogits = d loss / d logits * grad_out[0] (device scalar) as a dense [R][ldd]
 * matrix (the logits keep the grouped addressing); ignored rows get zeros.  loss.backward() of the two losses
 * above (transformer_word_level_bahdnau.py:230-232, seq2seqbasic.py:219-221). */
int icap_ce_loss_bwd_f32(const float* logits, int ld, int group_rows, int group_stride, const long long* targets,
                         int R, int V, long long ignore_index, const float* row_lse, const float* meta,
                         const float* grad_out, float* dlogits, int ldd, void* stream);

/* One AdamW step over a flat buffer of n floats (n % 4 == 0), torch.optim.AdamW
 * update order; step counts from 1; every gradient is multiplied by grad_scale first
 * (1/world_size after a summing all-reduce).  optimizer.step() of
 * transformer_word_level_bahdnau.py:191,233; transformer_word_level.py:138;
 * seq2seqbasic.py:205-210,222. */
int icap_adamw_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, void* stream);

/* icap_adamw_f32 with the step count t = *step_dev + step_bias read from device memory by the kernel: the form a
 * training step recorded in a HIP graph needs (launch arguments are frozen in a graph; the caller advances *step_dev
 * inside the graph).  Same update, same reference lines. */
int icap_adamw_dev_f32(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, const int* step_dev, int step_bias, float grad_scale, void* stream);

/* ------------------------------------------------------------------ training step: backward GEMMs
 * C[M][N] (+)= op(A) . B, B stored [K][N]; op(A) = A stored [M][K] (a_kmajor = 0: dX = dY . W, the input
 * gradient of every nn.Linear / 1x1 convolution) or A stored [K][M] (a_kmajor = 1: dW = dY^T . X, the weight
 * gradient, contraction over batch x positions).  ws: icap_gemm_bwd_ws_floats(M, N, K, accumulate) floats
 * (may be NULL when that is 0).  loss.backward() of transformer_word_level_bahdnau.py:232 through every
 * Linear of :22,60-66,98-100,122 and every 1x1 convolution of the EfficientNet of :14. */
int icap_gemm_bwd_f32(const float* a, int lda, int a_kmajor, const float* b, int ldb, float* c, int ldc, int M, int N,
                      int K, int accumulate, float* ws, void* stream);
long long icap_gemm_bwd_ws_floats(int M, int N, int K, int accumulate);

/* Column reductions over a [M][C] matrix, two passes in a fixed order.  mode 0: out1 = sum a (bias gradients,
 * BatchNorm mean with scale 1/M); 1: out1 = sum (a - v1[c])^2; 2: BatchNorm gradient sums out1 = sum g,
 * out2 = sum g * xhat with g = a * act'(x*v3 + v4), xhat = (x - v1) * v2; 3: LayerNorm parameter gradients
 * out1 = sum a, out2 = sum a * (x - r1[r]) * r2[r]; 4: out1 = sum (a - v1[c]), out2 = sum (a - v1[c])^2.  Results are multiplied by scale and optionally added onto
 * out.  ws: icap_col_reduce_ws_floats(M, C) floats.  nn.BatchNorm1d / BatchNorm2d in train mode
 * (transformer_word_level_bahdnau.py:64,123 and the EfficientNet blocks) and nn.LayerNorm (:57-58). */
int icap_col_reduce_f32(int mode, const float* a, int lda, const float* x, int ldx, const float* v1, const float* v2,
                        const float* v3, const float* v4, const float* r1, const float* r2, int act, int M, int C,
                        float scale, int accumulate, float* out1, float* out2, float* ws, void* stream);
long long icap_col_reduce_ws_floats(int M, int C);

/* BatchNorm in training mode: from the batch mean and the summed squared deviations compute rstd, the fused
 * normalisation y = x*s + b, and update the running statistics with `momentum` (unbiased variance). */
int icap_bn_finalize_f32(const float* mean, const float* sqdev, const float* gamma, const float* beta, float eps,
                         float momentum, int M, int C, float* rstd, float* s, float* b, float* run_mean,
                         float* run_var, void* stream);

/* The same from ONE pass over x: icap_col_reduce_f32 mode 4 gives s1 = sum (x - shift[c]) and s2 = sum (x - shift[c])^2
 * (shift: any vector near the column means, e.g. the first row of x); this call turns them into mean (written to
 * `mean`), rstd, s, b and the running-statistics update. */
int icap_bn_finalize_moments_f32(const float* shift, const float* s1, const float* s2, const float* gamma, const float* beta,
                                 float eps, float momentum, int M, int C, float* mean, float* rstd, float* s, float* b,
                                 float* run_mean, float* run_var, void* stream);

/* Both steps in two launches instead of three: the one-pass moments of x [M][C] (rows ldx apart) about its first row, and -
 * inside the reduction's second pass - mean, rstd, s, b and the running-statistics update.  ws: icap_col_reduce_ws_floats(M, C)
 * floats.  This is what nn.BatchNorm2d / BatchNorm1d do with a batch in train() mode (transformer_word_level_bahdnau.py:12-25,
 * 63-68 under model.train()). */
int icap_bn_train_stats_f32(const float* x, int ldx, const float* gamma, const float* beta, float eps, float momentum, int M, int C,
                            float* mean, float* rstd, float* s, float* b, float* run_mean, float* run_var, float* ws, void* stream);

/* Train-mode BatchNorm statistics WITHOUT a pass over the tensor: the kernel that produces a BatchNorm's input leaves, per
 * block of result rows, every column's sum and its squared deviations about the block's own mean (stat_part [P][2][C]), and
 * icap_bn_stats_from_parts_f32 merges the blocks (Chan's pairwise update, fixed order) into what icap_bn_train_stats_f32
 * returns - mean, rstd, s, b and the running-statistics update.  The producers (no scale / bias / residual / gate /
 * activation: a convolution in front of a BatchNorm has none, transformer_word_level_bahdnau.py:12-25 under model.train()):
 *   icap_conv2d_stats_f32        icap_conv2d_f32's kernel (any tile configuration); info[0] = P, info[1] = rows per block;
 *                                stat_part must hold icap_conv2d_stats_max_parts(M) * 2 * Cout floats (part_floats says so)
 *   icap_linear_f32x3_stats_f32  icap_linear_f32x3_f32's kernel (fp32 x, cfg as there); info as above
 *   icap_dwconv2d_stats_f32      icap_dwconv2d_f32's kernel; P = B * icap_dwconv_chunks(...), stat_cnt [P] = values per block
 * icap_bn_stats_from_parts_f32: cnt = stat_cnt, or null when block p holds min(rows_per_part, M - p * rows_per_part) rows;
 * ws: icap_bn_stats_from_parts_ws_floats(P, C) floats. */
long long icap_conv2d_stats_max_parts(long long M);
int icap_conv2d_stats_f32(const float* x, const float* w, float* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                          int pad_t, int pad_l, int Ho, int Wo, float* stat_part, long long part_floats, int* info, void* stream);
int icap_linear_f32x3_stats_f32(const float* x, const unsigned short* w3, float* y, int M, int K, int N, int cfg, float* stat_part,
                                long long part_floats, int* info, void* stream);
int icap_dwconv2d_stats_f32(const float* x, const float* w, float* y, int B, int H, int W, int C, int KS, int stride, int pad_t,
                            int pad_l, int Ho, int Wo, float* stat_part, float* stat_cnt, void* stream);
long long icap_bn_stats_from_parts_ws_floats(int P, int C);
int icap_bn_stats_from_parts_f32(const float* part, const float* cnt, int P, int rows_per_part, int M, int C, const float* gamma,
                                 const float* beta, float eps, float momentum, float* mean, float* rstd, float* s, float* b,
                                 float* run_mean, float* run_var, float* ws, void* stream);

/* y = act(x * s[c] + b[c]) (+ res): normalisation + activation (+ identity skip) over [M][C]. */
int icap_scale_bias_act_f32(const float* x, int ldx, const float* s, const float* b, const float* res, int ldr, int act,
                            int M, int C, float* y, int ldy, void* stream);

/* Gradient of y = act(BatchNorm(x)) with respect to x (mean == NULL: plain per-channel affine + activation). */
int icap_bn_act_bwd_f32(const float* dy, int ldd, const float* x, int ldx, const float* s, const float* b,
                        const float* mean, const float* rstd, const float* dgamma, const float* dbeta, int act, int M,
                        int C, float* dx, int ldo, void* stream);

/* nn.BatchNorm1d(V) applied to the logits of every time step of a teacher-forced pass (transformer_word_level_bahdnau.py:139:
 * self.fc holds Linear + BatchNorm1d and runs once per step, so each step is normalised with the statistics of its own B rows
 * and the running statistics receive one momentum update per step, in step order), all T steps per launch.
 * x [T*B][V] step-major (rows t*B ... t*B+B-1 = step t); y, dy [B][T][V]; mean, rstd, s, b, uvar, dgamma_t, dbeta_t [T][V]
 * (uvar: scratch for the unbiased variances).  Forward writes y and the per-step mean / rstd / fused scale s and bias b the
 * backward needs; backward returns dx [T*B][V] and the per-step parameter gradients (their sum over t is the layer's).
 * V % 4 == 0 and 16-byte aligned operands (error 2 otherwise).  Sums in a fixed order. */
int icap_bn_steps_fwd_f32(const float* x, const float* gamma, const float* beta, float eps, float momentum, int B, int T,
                          int V, float* y, float* mean, float* rstd, float* s, float* b, float* uvar, float* run_mean,
                          float* run_var, void* stream);
int icap_bn_steps_bwd_f32(const float* dy, const float* x, const float* mean, const float* rstd, const float* s, int B,
                          int T, int V, float* dgamma_t, float* dbeta_t, float* dx, void* stream);

/* y = alpha*x + beta*y (gradient accumulation where a tensor feeds several consumers). */
int icap_axpby_f32(const float* x, float alpha, float beta, float* y, long long n, void* stream);

/* nn.Dropout(p) in train mode (transformer_word_level_bahdnau.py:63,66,68,125) and, with group = elements per
 * sample, the EfficientNet drop-connect: y = x * mask, mask = 0 or 1/(1-p) from a counter-based generator keyed
 * by (seed, offset + index / group); mask [n / group] is kept for the backward pass. */
int icap_dropout_f32(const float* x, float p, unsigned long long seed, unsigned long long offset, long long n,
                     long long group, float* y, float* mask, void* stream);
/* icap_dropout_f32 with the generator offset = offset + *offset_base, the base read from device memory by the kernel
 * (for a training step recorded in a HIP graph: the caller advances the base inside the graph, so every replay draws
 * new masks).  Same reference lines. */
int icap_dropout_dev_f32(const float* x, float p, unsigned long long seed, unsigned long long offset,
                         const unsigned long long* offset_base, long long n, long long group, float* y, float* mask,
                         void* stream);
/* y = x * mask[i / group] (+ add): re-applies a stored mask (backward of dropout; forward with a given mask). */
int icap_mul_mask_f32(const float* x, const float* mask, long long group, const float* add, long long n, float* y,
                      void* stream);

/* nn.LayerNorm in a training step: forward keeps the normalised-input xs = (x (+ res)) * pre_scale, mean and
 * rstd per row; backward returns d/dx (times post_scale).  transformer_word_level_bahdnau.py:77,88. */
int icap_ln_train_fwd_f32(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                          float pre_scale, int R, int E, float* y, float* xs, float* mean, float* rstd, void* stream);
int icap_ln_bwd_f32(const float* dy, const float* xs, const float* gamma, const float* mean, const float* rstd,
                    float post_scale, int R, int E, float* dx, void* stream);

/* Gradient of one nn.LSTMCell step (transformer_word_level_bahdnau.py:136) with respect to the gate
 * pre-activations [B][4H] and the previous cell state, from dh and the dc arriving from step t+1. */
int icap_lstm_bwd_f32(const float* gates, const float* c_in, const float* c_out, const float* dh, const float* dc,
                      int B, int H, float* dgates, float* dc_prev, void* stream);

/* Gradient of one nn.GRU cell step (seq2seqbasic.py:52,63): from dh' to the pre-activation gradients of the input
 * half dgi and the recurrent half dgh ([B][3H], gate order r, z, n) and the direct path dh_direct = dh' * z. */
int icap_gru_bwd_f32(const float* gi, const float* gh, const float* h_in, const float* dh, int B, int H, float* dgi,
                     float* dgh, float* dh_direct, void* stream);

/* Gradient of Attention.forward (transformer_word_level_bahdnau.py:102-110) for one decode step: from dctx
 * [B][E] to dw [B][A] (query projections), du [B][A] and dfeat [B][E] (key side, optionally accumulated over
 * steps), per-query partials da_part [B][A], dab_part [B] of the scoring vector's gradient, dscore [B][B]. */
int icap_att_bwd_f32(const float* u, const float* w, const float* a_w, const float* alpha, const float* feat,
                     const float* dctx, int B, int A, int E, int accumulate, float* dscore, float* dw, float* da_part,
                     float* dab_part, float* du, float* dfeat, void* stream);

/* Gradient of the attention core of nn.MultiheadAttention (softmax(QK^T/sqrt(d))V per head, no mask, no attention
 * dropout) as the word / char transformer uses it: R query rows against Bk key rows (the images of the batch;
 * transformer_word_level.py:45,73-74).  q, dO, dq [R][E]; k, v, dk, dv [Bk][E]; ws: 2*R*heads*Bk floats. */
int icap_mha_bwd_f32(const float* q, const float* k, const float* v, const float* dO, int R, int Bk, int E, int heads,
                     float* dq, float* dk, float* dv, float* ws, void* stream);

/* nn.Embedding gradient: dtable[idx[p]] += dy[p], positions in order (deterministic); dtable pre-zeroed. */
int icap_embedding_bwd_f32(const long long* idx, const float* dy, int ldd, int P, int E, int V, float* dtable,
                           void* stream);

/* ------------------------------------------------------------------ training step: convolutional encoder
 * Gradients through the EfficientNet-B0 of get_cnn_model (transformer_word_level_bahdnau.py:12-25), which the
 * reference trains end to end (optimizer over caption_model.parameters(), :191).  NHWC fp32. */

/* y[b,p,c] = x[b,p,c] * g[b][c] + a[b][c] * alpha (x, g, a optional): squeeze-excite gate forward, its input
 * gradient together with the pooled mean's gradient, global-average-pool backward. */
int icap_chan_affine_f32(const float* x, const float* g, const float* a, float alpha, int B, int HW, int C, float* y,
                         void* stream);
/* out[b][c] = scale * sum_p x[b,p,c] * y[b,p,c] (y NULL: sum of x): gradient of the squeeze-excite gate, and
 * the pooled mean.  ws: icap_image_dot_ws_floats(B, HW, C) floats. */
int icap_image_dot_f32(const float* x, const float* y, int B, int HW, int C, float scale, float* out, float* ws,
                       void* stream);
long long icap_image_dot_ws_floats(int B, int HW, int C);
/* The BatchNorm apply in front of a squeeze-excite block together with the block's pooling (efficientnet_pytorch MBConvBlock:
 * _bn1 -> swish -> adaptive_avg_pool2d, train mode under transformer_word_level_bahdnau.py:12-25), one pass instead of two:
 * y[b,p,c] = act(x[b,p,c] * s[c] + b[c]), pooled[b][c] = scale * sum_p y[b,p,c].  y may be NULL (only the pooled means:
 * icap_act_gate_f32 then writes the gated activation straight from x).  C % 4 == 0, 16-byte aligned operands
 * (error 2 otherwise).  ws: icap_act_pool_ws_floats(B, HW, C) floats.  The sums are taken in a fixed order. */
int icap_act_pool_f32(const float* x, const float* s, const float* b, int act, int B, int HW, int C, float scale, float* y,
                      float* pooled, float* ws, void* stream);
long long icap_act_pool_ws_floats(int B, int HW, int C);
/* y[b,p,c] = act(x[b,p,c] * s[c] + b[c]) * g[b][c]: BatchNorm apply + activation + squeeze-excite gate from the BatchNorm's
 * input (same lines: _bn1 -> swish -> ... -> sigmoid gate -> multiply), so that the ungated activation is never stored.
 * icap_image_dot_act_f32 is icap_image_dot_f32 with its second operand recomputed the same way, act(y * s2[c] + b2[c])
 * (act 0 or swish) - the gate's gradient sum_p dy * swish(bn(z)) without the stored activation.  C % 4 == 0, 16-byte
 * aligned operands (error 2 otherwise). */
int icap_act_gate_f32(const float* x, const float* s, const float* b, int act, const float* g, int B, int HW, int C, float* y,
                      void* stream);
int icap_image_dot_act_f32(const float* x, const float* y, const float* s2, const float* b2, int act, int B, int HW, int C,
                           float scale, float* out, float* ws, void* stream);
/* Gradient of y = act(BatchNorm_train(x)) when the arriving gradient still has to pass the squeeze-excite gate (same lines,
 * backward): dy_eff[b,p,c] = dy[b,p,c] * gate[b][c] + dpool[b][c] * alpha is formed on the fly by both passes (the two column
 * sums dgamma / dbeta [C], then dx) and never written out.  x, dy, dx dense [B*HW][C]; gate, dpool [B][C]; s, b, mean, rstd:
 * what the forward pass kept.  C % 4 == 0 and 16-byte aligned operands (error 2 otherwise).
 * ws: icap_col_reduce_ws_floats(B*HW, C) floats. */
int icap_bn_act_bwd_se_f32(const float* dy, const float* x, const float* s, const float* b, const float* mean,
                           const float* rstd, const float* gate, const float* dpool, float alpha, int HW, int act, int M,
                           int C, float* dgamma, float* dbeta, float* dx, float* ws, void* stream);

/* The squeeze-excite branch of an MBConv block in training mode (efficientnet_pytorch MBConvBlock: _se_reduce -> swish ->
 * _se_expand -> sigmoid, under transformer_word_level_bahdnau.py:12-25 in model.train()), one launch forward and two backward
 * instead of six and eight tiny ones.  pooled [B][C] (per-image channel means), wr [S][C] = _se_reduce.weight, br [S],
 * we [C][S] = _se_expand.weight, be [C].  Forward: r_pre [B][S] = wr.pooled + br, e [B][C] = we.swish(r_pre) + be,
 * gate [B][C] = sigmoid(e).  Backward from dgate [B][C]: de [B][C], dr_pre [B][S] (scratch the caller provides), dpooled [B][C],
 * and the parameter gradients dwe [C][S], dbe [C], dwr [S][C], dbr [S] (overwritten; images summed in a fixed order).
 * C <= 2048, S <= 128. */
int icap_se_train_fwd_f32(const float* pooled, const float* wr, const float* br, const float* we, const float* be, float* r_pre,
                          float* e, float* gate, int B, int C, int S, void* stream);
int icap_se_train_bwd_f32(const float* dgate, const float* e, const float* r_pre, const float* pooled, const float* wr,
                          const float* we, float* de, float* dr_pre, float* dpooled, float* dwe, float* dbe, float* dwr, float* dbr,
                          int B, int C, int S, void* stream);

/* Depthwise convolution gradients (weights [K*K][C], TF-"same" padding given as pad_t / pad_l). */
int icap_dwconv_bwd_data_f32(const float* dy, const float* w, int B, int H, int W, int C, int K, int S, int pad_t,
                             int pad_l, int Ho, int Wo, float* dx, void* stream);
int icap_dwconv_bwd_weight_f32(const float* dy, const float* x, int B, int H, int W, int C, int K, int S, int pad_t,
                               int pad_l, int Ho, int Wo, float* dw, float* ws, void* stream);
long long icap_dwconv_bwd_weight_ws_floats(int B, int Ho, int Wo, int C, int K);

/* dst[c][r] = src[r][c] (src rows `ld_src` floats apart, dst dense [cols][rows]).  Used by the training step to hand a
 * Linear / 1x1-convolution weight to icap_conv2d_f32 with the contraction axis contiguous for the input gradient
 * dX = dY W (the backward of every nn.Linear / Conv2d on the path, e.g. transformer_word_level_bahdnau.py:188-196). */
int icap_transpose_f32(const float* src, int rows, int cols, int ld_src, float* dst, void* stream);

/* The random augmentations of the reference's image transform for caller-supplied per-image parameters, on decoded
 * uint8 pixels ahead of icap_resize_normalize_u8_f32: RandomHorizontalFlip, RandomRotation(20) (Pillow rotate: NEAREST, no
 * expansion, zero fill) and ColorJitter(contrast=0.3) (Pillow ImageEnhance.Contrast) - data.py:167-169,
 * datachar.py:125-127 - bit-exact against Pillow for the same parameters.  src, dst uint8 [B][H][W][3], dst != src;
 * params int32 [B][8] = {flip, a0, a1, a2, a3, a4, a5, 0} with the 16.16 fixed-point coefficients of Pillow's affine
 * pass (source x = (a2 + x*a0 + y*a1) >> 16, source y = (a5 + x*a3 + y*a4) >> 16); contrast fp32 [B] (1 = unchanged);
 * ws: icap_augment_ws_bytes(B) bytes, 8-byte aligned. */
long long icap_augment_ws_bytes(int B);
int icap_augment_u8(const unsigned char* src, int B, int H, int W, const int* params, const float* contrast, void* ws,
                    unsigned char* dst, void* stream);

/* Deterministic tail of the image transform (data.py:170-172, datachar.py:129-131: Resize((224, 224)) of a PIL image =
 * Pillow's BILINEAR resampler - separable triangle filter, 22-bit fixed-point taps, 8-bit intermediate after the
 * horizontal pass - then ToTensor and Normalize), bit-exact.  src [B][H][W][3] uint8 (decoded RGB, all B images of one
 * size); kx [Wo][ksx] / ky [Ho][ksy] int32 filter taps and bx [Wo][2] / by [Ho][2] (first input index, tap count) as
 * Pillow's precompute_coeffs + normalize_coeffs_8bpc give them for (W -> Wo) and (H -> Ho); mean_std: 3 means, 3 stds;
 * tmp: icap_resize_normalize_ws_bytes(B, H, Wo) bytes; out fp32, layout 0 = [B][3][Ho][Wo] (the reference's tensor),
 * 1 = [B][Ho][Wo][4] with a zero fourth channel.  The random flip / rotation / contrast jitter in front of the resize
 * (data.py:167-169) are not part of this call. */
long long icap_resize_normalize_ws_bytes(int B, int H, int Wo);
int icap_resize_normalize_u8_f32(const unsigned char* src, int B, int H, int W, const int* kx, const int* bx, int ksx,
                                 const int* ky, const int* by, int ksy, int Ho, int Wo, const float* mean_std, void* tmp,
                                 float* out, int layout, void* stream);

/* col[m][(r*KW+q)*C + c] = x[b, ho*S-pad_t+r, wo*S-pad_l+q, c]: patches of the stem convolution's input, so that
 * its weight gradient is one icap_gemm_bwd_f32 (a_kmajor = 1) call. */
int icap_im2col_f32(const float* x, int B, int H, int W, int C, int KH, int KW, int S, int pad_t, int pad_l, int Ho,
                    int Wo, float* col, void* stream);

/* Weight gradient of a convolution with few input channels, the 3-channel 3x3 stride-2 stem of get_cnn_model
 * (transformer_word_level_bahdnau.py:12-25, trained end to end), without a patch matrix: the patches are gathered from the
 * image inside the GEMM.  x [B][H][W][C] (C stored channels of which the first Cv are real), dz [B*Ho*Wo][Cout] (row stride
 * ldz); dw [Cout][(r*KW+q)*Cv + c].  Cout <= 32 and KH*KW*Cv <= 32 (error 3 otherwise: icap_im2col_f32 + icap_gemm_bwd_f32
 * cover the general case).  ws: icap_conv_wgrad_gather_ws_floats floats.  Partial sums are added in a fixed order. */
int icap_conv_wgrad_gather_f32(const float* dz, int ldz, const float* x, int B, int H, int W, int C, int Cv, int KH, int KW,
                               int S, int pad_t, int pad_l, int Ho, int Wo, int Cout, float* dw, float* ws, void* stream);
long long icap_conv_wgrad_gather_ws_floats(int B, int Ho, int Wo, int Cout, int N);

#ifdef __cplusplus
}
#endif
#endif /* ICAP_HIP_H */
