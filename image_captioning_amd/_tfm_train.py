"""Training-mode forward and hand-written backward of the word-level and char-level transformer captioners
(transformer_word_level.py:28-112, transformercharmiguel.py:30-113 under ``model.train()``).

The encoder block and the decoder block share one shape - attention (queries against the B images of the batch),
dropout, LayerNorm(residual), Linear-ReLU-dropout-Linear-dropout, LayerNorm(residual) - so both are the same
``AttentionBlockTrain`` autograd function: the encoder calls it with queries = keys = the image rows
(self-attention ACROSS the batch, line 45), the decoder with the caption embeddings as queries and the encoder
rows as keys (lines 73-74; tiling the keys T times does not change the softmax-weighted sums nor, summed over the
tiles, their gradients).  ``LinearTrain`` (output_fc) and ``EmbeddingTrain`` complete the chain.  The decoder's
``attention`` and ``layer_norm1`` are never called by the reference and get no gradient.
"""
from __future__ import annotations

import torch

from . import ops
from . import train_ops as T
from ._bah_train import _drop, _undrop
from .ops import ACT_RELU


class AttentionBlockTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xq, kv, in_w, in_b, out_w, out_b, lnA_g, lnA_b, lnB_g, lnB_b, w0, b0, w3, b3, heads, p, masks, same, eps):
        """xq [R, E] (queries and residual), kv [Bk, E] (keys = values); same: kv is xq (self-attention)."""
        E = xq.shape[1]
        xq = xq.float().contiguous()
        kv = xq if same else kv.float().contiguous()
        f = lambda t: t.detach().float().contiguous()
        iw, ib = in_w.detach().float(), in_b.detach().float()
        wq, wk, wv = iw[:E].contiguous(), iw[E:2 * E].contiguous(), iw[2 * E:].contiguous()
        q = ops.linear(xq, wq, ib[:E].contiguous())
        k = ops.linear(kv, wk, ib[E:2 * E].contiguous())
        v = ops.linear(kv, wv, ib[2 * E:].contiguous())
        R, Bk = xq.shape[0], kv.shape[0]
        core = ops.mha_core(q.view(R, 1, E), k.view(Bk, 1, E), v.view(Bk, 1, E), heads).view(R, E)
        att = ops.linear(core, f(out_w), f(out_b))
        m = masks or (None, None, None)
        att_d, m_att = _drop(att, p, m[0])
        x1, lnA_saved = T.layernorm_train(att_d, f(lnA_g), f(lnA_b), res=xq, eps=eps)
        h = ops.linear(x1, f(w0), f(b0), act=ACT_RELU)
        h_d, m_h = _drop(h, p, m[1])
        ff = ops.linear(h_d, f(w3), f(b3))
        ff_d, m_f = _drop(ff, p, m[2])
        out, lnB_saved = T.layernorm_train(ff_d, f(lnB_g), f(lnB_b), res=x1, eps=eps)
        ctx.saved = (xq, kv, wq, wk, wv, q, k, v, core, m_att, lnA_saved, x1, h, m_h, h_d, m_f, lnB_saved)
        ctx.params = (in_w, out_w, lnA_g, lnB_g, w0, w3)
        ctx.cfg = (heads, same, E)
        return out

    @staticmethod
    def backward(ctx, dout):
        xq, kv, wq, wk, wv, q, k, v, core, m_att, lnA_saved, x1, h, m_h, h_d, m_f, lnB_saved = ctx.saved
        in_w, out_w, lnA_g, lnB_g, w0, w3 = ctx.params
        heads, same, E = ctx.cfg
        f = lambda t: t.detach().float().contiguous()
        dsumB, dlnB_g, dlnB_b = T.layernorm_bwd(dout.float().contiguous(), lnB_saved, f(lnB_g))     # d(x1 + ff_d)
        dff = _undrop(dsumB, m_f)
        dw3 = T.gemm_tn(dff, h_d)
        db3 = T.col_sum(dff)
        dh_d = T.gemm_nn(dff, f(w3))
        dz = T.act_bwd(_undrop(dh_d, m_h), h, ACT_RELU)
        dw0 = T.gemm_tn(dz, x1)
        db0 = T.col_sum(dz)
        dx1 = T.gemm_nn(dz, f(w0))
        T.axpby(dsumB, dx1)                                                     # residual of the second LayerNorm
        dsumA, dlnA_g, dlnA_b = T.layernorm_bwd(dx1, lnA_saved, f(lnA_g))       # d(xq + att_d)
        datt = _undrop(dsumA, m_att)
        dout_w = T.gemm_tn(datt, core)
        dout_b = T.col_sum(datt)
        dcore = T.gemm_nn(datt, f(out_w))
        dq, dk, dv = T.mha_bwd(q, k, v, dcore, heads)
        din_w = torch.empty_like(in_w, dtype=torch.float32)
        din_b = torch.empty(3 * E, device=xq.device)
        T.gemm_tn(dq, xq, out=din_w[:E]); T.col_sum(dq, out=din_b[:E])
        T.gemm_tn(dk, kv, out=din_w[E:2 * E]); T.col_sum(dk, out=din_b[E:2 * E])
        T.gemm_tn(dv, kv, out=din_w[2 * E:]); T.col_sum(dv, out=din_b[2 * E:])
        dxq = T.gemm_nn(dq, wq)
        T.axpby(dsumA, dxq)                                                     # residual of the first LayerNorm
        dkv = T.gemm_nn(dk, wk)
        T.gemm_nn(dv, wv, out=dkv, accumulate=True)
        if same:
            T.axpby(dkv, dxq)
            dkv = None
        return (dxq, dkv, din_w, din_b, dout_w, dout_b, dlnA_g, dlnA_b, dlnB_g, dlnB_b, dw0, db0, dw3, db3, None, None, None, None, None)


class LinearTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x = x.float().contiguous()
        wk = w.detach().float().contiguous()
        ctx.saved = (x, wk)
        return ops.linear(x, wk, b.detach().float().contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, wk = ctx.saved
        dy = dy.float().contiguous()
        return T.gemm_nn(dy, wk), T.gemm_tn(dy, x), T.col_sum(dy)


class EmbeddingTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, idx):
        idx = idx.contiguous().view(-1)
        ctx.saved = (idx, table.shape[0])
        out, _ = ops.embedding(table.detach().float().contiguous(), idx)
        return out

    @staticmethod
    def backward(ctx, dy):
        idx, V = ctx.saved
        return T.embedding_bwd(idx, dy.float().contiguous(), V), None


def _block_masks(masks, keys, shape_rows, time_major=False):
    """Test masks -> per-row factor matrices.  Decoder masks arrive as [T, B, .] (the reference's layout) and are
    re-ordered to this module's (b, t) row order."""
    if masks is None:
        return None
    out = []
    for k in keys:
        m = masks.get(k)
        if m is not None:
            if time_major:
                m = m.permute(1, 0, 2)
            m = m.reshape(shape_rows, -1).float().contiguous()
        out.append(m)
    return tuple(out)


def encoder_train(mod, rows, masks=None):
    """rows [B, E] -> [B, E]: TransformerEncoderBlock.forward in train mode, attention across the batch."""
    a = mod.attention
    return AttentionBlockTrain.apply(rows, None, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                     mod.layer_norm1.weight, mod.layer_norm1.bias, mod.layer_norm2.weight, mod.layer_norm2.bias,
                                     mod.fc[0].weight, mod.fc[0].bias, mod.fc[3].weight, mod.fc[3].bias, a.num_heads, mod.dropout.p,
                                     _block_masks(masks, ("enc_att", "enc_fc1", "enc_fc2"), rows.shape[0]), True, mod.layer_norm1.eps)


def decoder_train(mod, x_rows, enc_rows, masks=None):
    """x_rows [R, E] caption embeddings, enc_rows [B, E] -> logits [R, V]: TransformerDecoderBlock.forward in train mode."""
    a = mod.cross_attention
    x = AttentionBlockTrain.apply(x_rows, enc_rows, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                                  mod.layer_norm2.weight, mod.layer_norm2.bias, mod.layer_norm3.weight, mod.layer_norm3.bias,
                                  mod.fc[0].weight, mod.fc[0].bias, mod.fc[3].weight, mod.fc[3].bias, a.num_heads, mod.dropout.p,
                                  _block_masks(masks, ("dec_att", "dec_fc1", "dec_fc2"), x_rows.shape[0], True), False, mod.layer_norm2.eps)
    return LinearTrain.apply(x, mod.output_fc.weight, mod.output_fc.bias)


def model_train_forward(model, img_embed, captions, masks=None):
    """ImageCaptioningModel.forward after the CNN, train mode: img_embed [B, E], captions [B, T] -> logits [B, T, V].
    masks (tests): {"enc_*": [B, .], "dec_*": [T, B, .]} factor tensors.  Every decoder row is independent of the
    others given the encoder rows, so the rows are kept in (b, t) order and the logits need no permute."""
    B, Tn = captions.shape
    enc = encoder_train(model.encoder, img_embed, masks)
    emb = EmbeddingTrain.apply(model.caption_embedding.weight, captions)               # rows (b, t)
    logits = decoder_train(model.decoder, emb, enc, masks)                             # [B*T, V]
    return logits.view(B, Tn, -1)
