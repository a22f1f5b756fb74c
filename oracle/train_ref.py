"""TEST INFRASTRUCTURE - CPU restatement of the reference's teacher-forced TRAINING forward (train-mode
BatchNorm, dropout with caller-supplied masks) for the Bahdanau captioner; gradients come from torch autograd.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Pinned by tests/golden/bahdanau_train_small.npz, produced by oracle/make_golden_train.py from the reference's own
classes in train() mode with dropout = 0 (the only setting in which the reference is deterministic).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _drop(x, mask):
    """mask: None (no dropout) or a tensor of factors (0 or 1/(1-p)) shaped like x."""
    return x if mask is None else x * mask


def bah_encoder_train(sd, rows, bn_buf, masks=None, prefix="encoder."):
    """transformer_word_level_bahdnau.py:70-90 in train mode.  rows [B, E] -> [B, E] (the reference returns
    [1, B, E]).  bn_buf: dict with 'running_mean', 'running_var' tensors updated in place (momentum 0.1)."""
    m = masks or {}
    p = lambda k: sd[prefix + k]
    E = rows.shape[1]
    wv, bv = p("attention.in_proj_weight")[2 * E:], p("attention.in_proj_bias")[2 * E:]
    att = F.linear(F.linear(rows, wv, bv), p("attention.out_proj.weight"), p("attention.out_proj.bias"))   # one key: softmax = 1
    x = F.layer_norm(rows + _drop(att, m.get("enc_att")), (E,), p("layer_norm1.weight"), p("layer_norm1.bias"))
    h = _drop(F.relu(F.linear(x, p("fc.0.weight"), p("fc.0.bias"))), m.get("enc_fc1"))
    h = F.batch_norm(h, bn_buf["running_mean"], bn_buf["running_var"], p("fc.3.weight"), p("fc.3.bias"), training=True, momentum=0.1, eps=1e-5)
    f = _drop(F.linear(h, p("fc.4.weight"), p("fc.4.bias")), m.get("enc_fc2"))
    return F.layer_norm(f + f, (E,), p("layer_norm2.weight"), p("layer_norm2.bias"))                          # lines 85-88


def bah_decoder_train(sd, captions, feats, bn_buf, masks=None, prefix="decoder."):
    """transformer_word_level_bahdnau.py:127-141 in train mode.  feats [B, ENC] -> logits [B, T, V]."""
    m = masks or {}
    p = lambda k: sd[prefix + k]
    B, T = captions.shape
    H = p("lstm.weight_hh").shape[1]
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    u = F.linear(feats, p("attention.U.weight"), p("attention.U.bias"))                  # [B(j), A]
    outs = []
    for t in range(T):
        emb = p("embedding.weight")[captions[:, t]]
        w = F.linear(h, p("attention.W.weight"), p("attention.W.bias"))                  # [B(i), A]
        score = F.linear(torch.tanh(u.unsqueeze(0) + w.unsqueeze(1)), p("attention.A.weight"), p("attention.A.bias")).squeeze(2)
        alpha = torch.softmax(score, dim=1)
        ctx = alpha @ feats
        gates = F.linear(torch.cat([emb, ctx], 1), p("lstm.weight_ih"), p("lstm.bias_ih")) + F.linear(h, p("lstm.weight_hh"), p("lstm.bias_hh"))
        i, f, g, o = gates.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        hd = _drop(h, None if m.get("dec_out") is None else m["dec_out"][:, t])
        logit = F.linear(hd, p("fc.0.weight"), p("fc.0.bias"))
        logit = F.batch_norm(logit, bn_buf["running_mean"], bn_buf["running_var"], p("fc.1.weight"), p("fc.1.bias"), training=True,
                             momentum=0.1, eps=1e-5)
        outs.append(logit.unsqueeze(1))
    return torch.cat(outs, 1)


def bah_train_forward(sd, img_features, captions, enc_bn, dec_bn, masks=None):
    """ImageCaptioningModel.forward (lines 156-166) from post-CNN features, train mode."""
    enc = bah_encoder_train(sd, img_features, enc_bn, masks)
    return bah_decoder_train(sd, captions, enc, dec_bn, masks)


# ---------------------------------------------------------------------------- word / char transformer, train mode
def tfm_train_forward(sd, img_embed, captions, enc_heads, dec_heads, masks=None):
    """ImageCaptioningModel.forward after the CNN (transformer_word_level.py:97-112, transformercharmiguel.py:98-113)
    in train mode: the eval computation of oracle.torch_ref plus the five dropouts of lines 37, 40, 42 (encoder) and
    67, 69, 71 (decoder).  masks: {"enc_att" [B, E], "enc_fc1" [B, FF], "enc_fc2" [B, E], "dec_att" [T, B, E],
    "dec_fc1" [T, B, FF], "dec_fc2" [T, B, E]} factor tensors; a missing key means no dropping.
    Pinned by tests/golden/{word,char}_tfm_train_small.npz (reference classes, dropout 0)."""
    from oracle.torch_ref import _lin, _ln, mha
    m = masks or {}
    B = img_embed.shape[0]
    # encoder: after the permute the B images are the attention sequence (L = B, N = 1)
    x = img_embed.unsqueeze(0).permute(1, 0, 2)
    a = mha(sd, "encoder.attention", x, x, x, enc_heads)
    x = _ln(sd, "encoder.layer_norm1", x + _drop(a, None if m.get("enc_att") is None else m["enc_att"].view(B, 1, -1)))
    h = _drop(torch.relu(_lin(sd, "encoder.fc.0", x)), None if m.get("enc_fc1") is None else m["enc_fc1"].view(B, 1, -1))
    f = _drop(_lin(sd, "encoder.fc.3", h), None if m.get("enc_fc2") is None else m["enc_fc2"].view(B, 1, -1))
    enc = _ln(sd, "encoder.layer_norm2", x + f)                                         # [B, 1, E]
    # decoder
    y = sd["caption_embedding.weight"][captions].permute(1, 0, 2)                       # [T, B, E]
    T = y.shape[0]
    kv = enc.repeat(T, B, 1)
    ca = mha(sd, "decoder.cross_attention", y, kv, kv, dec_heads)
    y = _ln(sd, "decoder.layer_norm2", y + _drop(ca, m.get("dec_att")))
    h = _drop(torch.relu(_lin(sd, "decoder.fc.0", y)), m.get("dec_fc1"))
    f = _drop(_lin(sd, "decoder.fc.3", h), m.get("dec_fc2"))
    y = _ln(sd, "decoder.layer_norm3", y + f)
    return _lin(sd, "decoder.output_fc", y).permute(1, 0, 2)
