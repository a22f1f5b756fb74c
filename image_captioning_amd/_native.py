"""ctypes binding of ``libicap_hip.so`` (C ABI declared in ``include/icap_hip.h``).

There is no fallback: if the library is missing or a symbol cannot be resolved the
import of any compute path fails with a clear error, and every non-zero return
code from the library is raised as :class:`IcapError`.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libicap_hip.so")

ABI_VERSION = 1

ACT_NONE, ACT_RELU, ACT_SWISH, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4


class IcapError(RuntimeError):
    pass


class RnnWeights(C.Structure):
    _fields_ = [
        ("layers", C.c_int), ("H", C.c_int), ("I", C.c_int), ("V", C.c_int), ("is_lstm", C.c_int),
        ("w_ih", C.c_void_p * 4), ("w_hh", C.c_void_p * 4), ("b_ih", C.c_void_p * 4), ("b_hh", C.c_void_p * 4),
        ("fc_w", C.c_void_p), ("fc_b", C.c_void_p), ("in_table", C.c_void_p),
    ]


class TfmWeights(C.Structure):
    _fields_ = [
        ("E", C.c_int), ("FF", C.c_int), ("V", C.c_int),
        ("emb", C.c_void_p),
        ("ln2_g", C.c_void_p), ("ln2_b", C.c_void_p), ("ln3_g", C.c_void_p), ("ln3_b", C.c_void_p),
        ("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p), ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p),
        ("out_w", C.c_void_p), ("out_b", C.c_void_p),
    ]


class BahWeights(C.Structure):
    _fields_ = [
        ("E", C.c_int), ("H", C.c_int), ("V", C.c_int),
        ("emb", C.c_void_p),
        ("w_ih", C.c_void_p), ("w_hh", C.c_void_p), ("b_ih", C.c_void_p), ("b_hh", C.c_void_p),
        ("fc_w", C.c_void_p), ("fc_scale", C.c_void_p), ("fc_bias", C.c_void_p),
    ]


_p, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> (restype, argtypes); must list every function include/icap_hip.h declares.
SIGNATURES = {
    "icap_abi_version": (_i, []),
    "icap_last_error": (C.c_char_p, []),
    "icap_nchw_to_nhwc_pad_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "icap_conv2d_f32": (_i, [_p] * 7 + [_i] * 13 + [_p]),
    "icap_conv2d_tune_f32": (_i, [_p] * 7 + [_i] * 13 + [_p]),
    "icap_conv2d_tuned_config": (_i, [_i] * 10),
    "icap_conv2d_num_configs": (_i, []),
    "icap_conv2d_use_tuned": (_i, [_i]),
    "icap_conv2d_cfg_f32": (_i, [_p] * 7 + [_i] * 14 + [_p]),
    "icap_linear_argmax_f32": (_i, [_p] * 7 + [_i, _i, _i, _p]),
    "icap_linear_argmax_parts": (_i, [_i]),
    "icap_next_token_f32": (_i, [_p, _p, _i, _p, _i, _p, _p, _p, _p, _f, _p, _i, _i, _i, _p]),
    "icap_dwconv2d_f32": (_i, [_p] * 6 + [_i] * 11 + [_p]),
    "icap_dwconv_chunks": (_i, [_i] * 5),
    "icap_expand_dwconv_f32": (_i, [_p] * 9 + [_i] * 11 + [_p]),
    "icap_expand_dwconv_chunks": (_i, [_i] * 6),
    "icap_expand_dwconv_whole_image": (_i, [_i] * 6),
    "icap_expand_dwconv_f32x3_f32": (_i, [_p] * 9 + [_i] * 11 + [_p]),
    "icap_expand_dwconv_img_x3_we_bytes": (_ll, [_i] * 5),
    "icap_expand_dwconv_img_x3_pack": (_i, [_p, _p] + [_i] * 5 + [_p]),
    "icap_expand_dwconv_img_x3_f32": (_i, [_p] * 9 + [_i] * 8 + [_p]),
    "icap_expand_dwconv_img_x3_s2_supported": (_i, [_i] * 7),
    "icap_expand_dwconv_img_x3_s2_f32": (_i, [_p] * 9 + [_i] * 8 + [_p]),
    "icap_stem_dwconv_f32": (_i, [_p] * 9 + [_i] * 12 + [_p]),
    "icap_stem_dwconv_nchw_f32": (_i, [_p] * 9 + [_i] * 12 + [_p]),
    "icap_stem_dwconv_valu_f32": (_i, [_p] * 9 + [_i] * 12 + [_p]),
    "icap_stem_dwconv_valu_supported": (_i, [_i]),
    "icap_stem_dwconv_chunks": (_i, [_i] * 3),
    "icap_expand_dwconv_bf16_f32": (_i, [_p] * 8 + [_i, _p] + [_i] * 11 + [_p]),
    "icap_stem_dwconv_bf16_f32": (_i, [_p] * 8 + [_i, _p] + [_i] * 12 + [_p]),
    "icap_se_gate_f32": (_i, [_p, _i, _f, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "icap_maxpool2d_f32": (_i, [_p, _p] + [_i] * 9 + [_p]),
    "icap_adaptive_avgpool2d_f32": (_i, [_p, _p] + [_i] * 6 + [_p]),
    "icap_embedding_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "icap_layernorm_f32": (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _f, _f, _p]),
    "icap_layernorm_split3_f32": (_i, [_p, _p, _i, _p, _p, _p, _p, _i, _i, _f, _f, _p]),
    "icap_argmax_rows_f32": (_i, [_p, _i, _p, _i, _i, _i, _p, _p]),
    "icap_linear_bf16_f32": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "icap_header_hash": (C.c_uint, []),
    "icap_to_bf16_f32": (_i, [_p, _p, _ll, _p]),
    "icap_linear_x3dl_f32": (_i, [_p] * 7 + [_i] * 5 + [_p]),
    "icap_linear_x3dl_argmax_f32": (_i, [_p] * 7 + [_i] * 3 + [_p]),
    "icap_split3_f32": (_i, [_p, _p, _ll, _p]),
    "icap_linear_f32x3_f32": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "icap_linear_f32x3_argmax_f32": (_i, [_p, _i] + [_p] * 6 + [_i, _i, _i, _p]),
    "icap_linear_bf16_argmax_f32": (_i, [_p] * 7 + [_i, _i, _i, _p]),
    "icap_linear_bf16_argmax_parts": (_i, [_i]),
    "icap_topk_sample_f32": (_i, [_p, _i, _i, _i, _i, C.c_ulonglong, C.c_ulonglong, _p, _p, _p, _i, _p]),
    "icap_mha_core_f32": (_i, [_p, _p, _p, _p] + [_i] * 6 + [_p]),
    "icap_bahdanau_attention_f32": (_i, [_p, _p, _p, _f, _p, _p, _p, _i, _i, _i, _p]),
    "icap_lstm_pointwise_f32": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "icap_gru_pointwise_f32": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "icap_gemv_f32": (_i, [_p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "icap_gemv_blocks": (_i, [_i]),
    "icap_argmax_finalize_f32": (_i, [_p, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _p]),
    "icap_lstm_cell_f32": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p]),
    "icap_gru_cell_f32": (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p]),
    "icap_greedy_rnn_f32": (_i, [C.POINTER(RnnWeights), _p, _p, _p, _p, _i, _i, _p]),
    "icap_greedy_rnn_ws_floats": (_ll, [_i] * 5),
    "icap_greedy_rnn_static_ws_floats": (_i, []),
    "icap_greedy_rnn_static_f32": (_i, [C.POINTER(RnnWeights), _p, _p, _p, _i, _i, _p]),
    "icap_greedy_tfm_f32": (_i, [C.POINTER(TfmWeights), _p, _p, _p, _p, _p, _i, _i, _p]),
    "icap_greedy_tfm_ws_floats": (_ll, [_i] * 4),
    "icap_table_walk": (_i, [_p, _p, _p, _p, _i, _p, _i, _i, _i, _p]),
    "icap_greedy_tfm_persist_max_nb": (_i, [_i] * 3),
    "icap_greedy_tfm_persist_ws_bytes": (_ll, [_i] * 4),
    "icap_greedy_tfm_persist_f32": (_i, [C.POINTER(TfmWeights), _p, _p, _p, _p, _p, _i, _i, _p]),
    "icap_greedy_bah_f32": (_i, [C.POINTER(BahWeights), _p, _i, _p, _p, _p, _p, _i, _i, _p]),
    "icap_greedy_rnn_w16_f32": (_i, [C.POINTER(RnnWeights), _p, _p, _p, _p, _i, _i, _p]),
    "icap_greedy_tfm_w16_f32": (_i, [C.POINTER(TfmWeights), _p, _p, _p, _p, _p, _i, _i, _p]),
    "icap_greedy_bah_w16_f32": (_i, [C.POINTER(BahWeights), _p, _i, _p, _p, _p, _p, _i, _i, _p]),
    "icap_greedy_bah_ws_floats": (_ll, [_i] * 5),
    "icap_linear_thin_supported": (_i, [_i] * 2),
    "icap_linear_f32x3_tall_supported": (_i, [_i] * 4),
    "icap_linear_f32x3_splitk_ws_floats": (_ll, [_i] * 3),
    "icap_linear_f32x3_splitk_f32": (_i, [_p] * 6 + [_i, _p, _p, _ll] + [_i] * 6 + [_p]),
    "icap_linear_thin_f32": (_i, [_p] * 6 + [_i, _p, _ll, _i, _i, _p]),
    "icap_greedy_bah_persist_max_nb": (_i, [_i] * 2),
    "icap_greedy_bah_persist_ws_bytes": (_ll, [_i] * 3),
    "icap_greedy_bah_persist_f32": (_i, [_p] * 10 + [_i] * 4 + [_p]),
    "icap_greedy_bah_fused_ws_floats": (_ll, [_i] * 3),
    "icap_greedy_bah_fused_f32": (_i, [_p] * 10 + [_i] * 4 + [_p]),
    "icap_greedy_rnn2_persist_max_nb": (_i, [_i] * 3),
    "icap_greedy_rnn2_persist_ws_bytes": (_ll, [_i] * 4),
    "icap_greedy_rnn2_persist_f32": (_i, [_i] + [_p] * 13 + [_i] * 4 + [_p]),
    "icap_greedy_rnn2_fused_ws_floats": (_ll, [_i] * 4),
    "icap_greedy_rnn2_fused_f32": (_i, [_i] + [_p] * 13 + [_i] * 4 + [_p]),
    "icap_ce_loss_fwd_f32": (_i, [_p, _i, _i, _i, _p, _i, _i, _ll, _i, _p, _p, _p, _p]),
    "icap_ce_loss_bwd_f32": (_i, [_p, _i, _i, _i, _p, _i, _i, _ll, _p, _p, _p, _p, _i, _p]),
    "icap_adamw_f32": (_i, [_p, _p, _p, _p, _ll, _f, _f, _f, _f, _f, _i, _f, _p]),
    "icap_adamw_dev_f32": (_i, [_p, _p, _p, _p, _ll, _f, _f, _f, _f, _f, _p, _i, _f, _p]),
    "icap_gemm_bwd_f32": (_i, [_p, _i, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p]),
    "icap_gemm_bwd_ws_floats": (_ll, [_i, _i, _i, _i]),
    "icap_col_reduce_f32": (_i, [_i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p, _p, _p, _p]),
    "icap_col_reduce_ws_floats": (_ll, [_i, _i]),
    "icap_bn_finalize_f32": (_i, [_p, _p, _p, _p, _f, _f, _i, _i, _p, _p, _p, _p, _p, _p]),
    "icap_bn_train_stats_f32": (_i, [_p, _i, _p, _p, _f, _f, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "icap_conv2d_stats_max_parts": (_ll, [_ll]),
    "icap_conv2d_stats_f32": (_i, [_p, _p, _p] + [_i] * 12 + [_p, _ll, _p, _p]),
    "icap_linear_f32x3_stats_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _ll, _p, _p]),
    "icap_dwconv2d_stats_f32": (_i, [_p, _p, _p] + [_i] * 10 + [_p, _p, _p]),
    "icap_bn_stats_from_parts_ws_floats": (_ll, [_i, _i]),
    "icap_bn_stats_from_parts_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p]),
    "icap_bn_finalize_moments_f32": (_i, [_p, _p, _p, _p, _p, _f, _f, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "icap_scale_bias_act_f32": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "icap_bn_act_bwd_f32": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _p]),
    "icap_axpby_f32": (_i, [_p, _f, _f, _p, _ll, _p]),
    "icap_dropout_f32": (_i, [_p, _f, C.c_ulonglong, C.c_ulonglong, _ll, _ll, _p, _p, _p]),
    "icap_dropout_dev_f32": (_i, [_p, _f, C.c_ulonglong, C.c_ulonglong, _p, _ll, _ll, _p, _p, _p]),
    "icap_mul_mask_f32": (_i, [_p, _p, _ll, _p, _ll, _p, _p]),
    "icap_ln_train_fwd_f32": (_i, [_p, _p, _p, _p, _f, _f, _i, _i, _p, _p, _p, _p, _p]),
    "icap_ln_bwd_f32": (_i, [_p, _p, _p, _p, _p, _f, _i, _i, _p, _p]),
    "icap_lstm_bwd_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "icap_gru_bwd_f32": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p]),
    "icap_att_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "icap_mha_bwd_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "icap_embedding_bwd_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "icap_chan_affine_f32": (_i, [_p, _p, _p, _f, _i, _i, _i, _p, _p]),
    "icap_image_dot_f32": (_i, [_p, _p, _i, _i, _i, _f, _p, _p, _p]),
    "icap_image_dot_ws_floats": (_ll, [_i, _i, _i]),
    "icap_bn_steps_fwd_f32": (_i, [_p, _p, _p, _f, _f, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "icap_bn_steps_bwd_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "icap_act_gate_f32": (_i, [_p, _p, _p, _i, _p, _i, _i, _i, _p, _p]),
    "icap_image_dot_act_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p]),
    "icap_act_pool_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p]),
    "icap_act_pool_ws_floats": (_ll, [_i, _i, _i]),
    "icap_bn_act_bwd_se_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _f, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "icap_se_train_fwd_f32": (_i, [_p] * 8 + [_i, _i, _i, _p]),
    "icap_se_train_bwd_f32": (_i, [_p] * 13 + [_i, _i, _i, _p]),
    "icap_dwconv_bwd_data_f32": (_i, [_p, _p] + [_i] * 10 + [_p, _p]),
    "icap_dwconv_bwd_weight_f32": (_i, [_p, _p] + [_i] * 10 + [_p, _p, _p]),
    "icap_dwconv_bwd_weight_ws_floats": (_ll, [_i] * 5),
    "icap_transpose_f32": (_i, [_p, _i, _i, _i, _p, _p]),
    "icap_conv2d_bf16_f32": (_i, [_p] * 6 + [_i] * 15 + [_p]),
    "icap_maxpool2d_bf16": (_i, [_p, _p] + [_i] * 9 + [_p]),
    "icap_adaptive_avgpool2d_bf16_f32": (_i, [_p, _p] + [_i] * 6 + [_p]),
    "icap_resize_normalize_ws_bytes": (C.c_longlong, [_i, _i, _i]),
    "icap_augment_ws_bytes": (C.c_longlong, [_i]),
    "icap_augment_u8": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "icap_resize_normalize_u8_f32": (_i, [_p, _i, _i, _i, _p, _p, _i, _p, _p, _i, _i, _i, _p, _p, _p, _i, _p]),
    "icap_conv_wgrad_gather_f32": (_i, [_p, _i, _p] + [_i] * 13 + [_p, _p, _p]),
    "icap_conv_wgrad_gather_ws_floats": (_ll, [_i, _i, _i, _i, _i]),
    "icap_im2col_f32": (_i, [_p] + [_i] * 11 + [_p, _p]),
}

_lib = None


HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "icap_hip.h")


def header_hash():
    """POSIX cksum CRC of include/icap_hip.h (what the Makefile bakes into the library), None when the header is not there."""
    try:
        data = open(HEADER_PATH, "rb").read()
    except OSError:
        return None
    crc = 0
    table = _cksum_table()
    for b in data:
        crc = ((crc << 8) & 0xFFFFFFFF) ^ table[((crc >> 24) ^ b) & 0xFF]
    n = len(data)
    while n:
        crc = ((crc << 8) & 0xFFFFFFFF) ^ table[((crc >> 24) ^ (n & 0xFF)) & 0xFF]
        n >>= 8
    return (~crc) & 0xFFFFFFFF


def _cksum_table():
    t = []
    for i in range(256):
        c = i << 24
        for _ in range(8):
            c = ((c << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if c & 0x80000000 else (c << 1) & 0xFFFFFFFF
        t.append(c)
    return t


def load():
    """Load the shared library once; raise IcapError when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IcapError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C image_captioning_amd/csrc`; there is no CPU fallback")
    # torch first: its own copy of the HIP runtime must be the one in the process before this library asks the loader
    # for libamdhip64 - loaded the other way round, the process ends up with two runtimes and the second one to
    # initialise finds no device
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover - build error
            raise IcapError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.icap_abi_version() != ABI_VERSION:
        raise IcapError(f"ABI mismatch: library {lib.icap_abi_version()} vs binding {ABI_VERSION}")
    built, here = lib.icap_header_hash(), header_hash()
    if built and here is not None and built != here:
        raise IcapError(f"{LIB_PATH} was built against another version of include/icap_hip.h (cksum {built}, the header here has "
                        f"{here}): rebuild it with `make -C image_captioning_amd/csrc`")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().icap_last_error().decode("utf-8", "replace")
        raise IcapError(f"{what or 'icap'} failed (code {rc}): {msg}")
