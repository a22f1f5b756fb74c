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
