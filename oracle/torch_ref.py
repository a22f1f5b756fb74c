"""CPU oracle for the decoder side of the captioning hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``image_captioning_amd/`` imports this
file; it is used by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` as the checker / timed baseline, never as
part of the product path.

It restates, as plain functions over a ``state_dict`` (fp32, CPU, eval mode), the
arithmetic of the reference's model classes.  Each function cites the reference
lines it follows.  The restatement is pinned against the reference itself:
``oracle/make_golden.py`` imports the reference's classes from ``/root/reference``
(third-party imports stubbed), loads the same seeded ``state_dict`` into them, and
stores their outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks this file against those vectors.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- primitives


def _lin(sd, p, x):
    return x @ sd[p + ".weight"].t() + sd[p + ".bias"]


def _ln(sd, p, x, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * sd[p + ".weight"] + sd[p + ".bias"]


def _bn_eval(sd, p, x, eps=1e-5):
    return (x - sd[p + ".running_mean"]) / torch.sqrt(sd[p + ".running_var"] + eps) * sd[p + ".weight"] + sd[p + ".bias"]


def mha(sd, p, q, k, v, heads):
    """torch.nn.MultiheadAttention forward, sequence-first [L, N, E], no masks, eval mode."""
    E = q.shape[-1]
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    Q = q @ w[:E].t() + b[:E]
    K = k @ w[E:2 * E].t() + b[E:2 * E]
    V = v @ w[2 * E:].t() + b[2 * E:]
    Lq, N, _ = Q.shape
    Lk = K.shape[0]
    hd = E // heads
    Q = Q.reshape(Lq, N * heads, hd).transpose(0, 1) / math.sqrt(hd)
    K = K.reshape(Lk, N * heads, hd).transpose(0, 1)
    V = V.reshape(Lk, N * heads, hd).transpose(0, 1)
    att = torch.softmax(Q @ K.transpose(1, 2), dim=-1) @ V            # [N*heads, Lq, hd]
    att = att.transpose(0, 1).reshape(Lq, N, E)
    return att @ sd[p + ".out_proj.weight"].t() + sd[p + ".out_proj.bias"]


def lstm_cell(w_ih, w_hh, b_ih, b_hh, x, h, c):
    """Gate order i, f, g, o (torch.nn.LSTM / LSTMCell)."""
    g = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    H = h.shape[-1]
    i, f, gg, o = g[..., :H], g[..., H:2 * H], g[..., 2 * H:3 * H], g[..., 3 * H:]
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def gru_cell(w_ih, w_hh, b_ih, b_hh, x, h):
    """Gate order r, z, n (torch.nn.GRU)."""
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    H = h.shape[-1]
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1 - z) * n + z * h


def top2_gap(logits):
    t = torch.topk(logits, 2, dim=-1).values
    return (t[..., 0] - t[..., 1])


# --------------------------------------------------------------------------- seq2seq (seq2seqbasic.py)


def _rnn_layers(sd, p="rnn"):
    L = 0
    while f"{p}.weight_ih_l{L}" in sd:
        L += 1
    return L


def _rnn_step(sd, kind, x, state, p="rnn"):
    """One time step through the stacked RNN.  state: list of (h, c) or (h,) per layer."""
    new = []
    for l, st in enumerate(state):
        w = [sd[f"{p}.{n}_l{l}"] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        if kind == "lstm":
            h, c = lstm_cell(*w, x, *st)
            new.append((h, c))
        else:
            h = gru_cell(*w, x, st[0])
            new.append((h,))
        x = h
    return x, new


def _zero_state(sd, kind, B, p="rnn"):
    H = sd[f"{p}.weight_hh_l0"].shape[1]
    z = lambda: torch.zeros(B, H)
    return [(z(), z()) if kind == "lstm" else (z(),) for _ in range(_rnn_layers(sd, p))]


def rnn_kind(sd, p="rnn"):
    H = sd[f"{p}.weight_hh_l0"].shape[1]
    return "lstm" if sd[f"{p}.weight_hh_l0"].shape[0] == 4 * H else "gru"


def seq2seq_encoder_head(sd, feats):
    """EncoderCNN.forward after the backbone: fc then eval BatchNorm1d (seq2seqbasic.py:43-44)."""
    return _bn_eval(sd, "bn", _lin(sd, "fc", feats))


def seq2seq_decoder_forward(sd, features, captions):
    """DecoderRNN.forward (seq2seqbasic.py:59-66): features [B, E], captions [B, T] -> [B, T, V]."""
    kind = rnn_kind(sd)
    emb = sd["embed.weight"][captions[:, :-1]]
    inputs = _lin(sd, "feature_fc", torch.cat((features.unsqueeze(1), emb), dim=1))
    state = _zero_state(sd, kind, features.shape[0])
    outs = []
    for t in range(inputs.shape[1]):
        h, state = _rnn_step(sd, kind, inputs[:, t], state)
        outs.append(h)
    return _lin(sd, "fc", torch.stack(outs, 1))


def seq2seq_sample(sd, feature, max_len=20, end_idx=3, stop_at_end=True):
    """DecoderRNN.sample (seq2seqbasic.py:68-91): feature [E] -> (ids, logits [steps, V]).
    The image feature is the first RNN input; the start token is never fed; the hidden state starts at zero."""
    kind = rnn_kind(sd)
    x = _lin(sd, "feature_fc", feature.unsqueeze(0))
    state = _zero_state(sd, kind, 1)
    ids, logits = [], []
    for _ in range(max_len):
        h, state = _rnn_step(sd, kind, x, state)
        out = _lin(sd, "fc", h)
        tok = int(out.argmax(1))
        ids.append(tok)
        logits.append(out[0])
        if stop_at_end and tok == end_idx:
            break
        x = _lin(sd, "feature_fc", sd["embed.weight"][tok].unsqueeze(0))
    return ids, torch.stack(logits)


# --------------------------------------------------------------------------- word / char transformer


def tfm_encoder(sd, img_embed, heads, p="encoder"):
    """TransformerEncoderBlock.forward (transformer_word_level.py:43-49) on the model's [1, B, E] input:
    after the permute the B images form the attention sequence.  Returns [B, 1, E]."""
    x = img_embed.unsqueeze(0).permute(1, 0, 2)                        # [B, 1, E] = (L=B, N=1, E)
    a = mha(sd, p + ".attention", x, x, x, heads)
    x = _ln(sd, p + ".layer_norm1", x + a)
    f = _lin(sd, p + ".fc.3", torch.relu(_lin(sd, p + ".fc.0", x)))
    return _ln(sd, p + ".layer_norm2", x + f)


def tfm_decoder(sd, x, encoder_out, heads, p="decoder"):
    """TransformerDecoderBlock.forward (transformer_word_level.py:72-79): x [T, B, E], encoder_out [B, 1, E].
    Keys/values are the encoder rows tiled to [B*T, B, E]: every batch column sees all B images."""
    T, B, _ = x.shape
    kv = encoder_out.repeat(T, B, 1)
    ca = mha(sd, p + ".cross_attention", x, kv, kv, heads)
    x = _ln(sd, p + ".layer_norm2", x + ca)
    f = _lin(sd, p + ".fc.3", torch.relu(_lin(sd, p + ".fc.0", x)))
    x = _ln(sd, p + ".layer_norm3", x + f)
    return _lin(sd, p + ".output_fc", x)


def tfm_forward_features(sd, img_embed, captions, enc_heads, dec_heads):
    """ImageCaptioningModel.forward after the CNN (transformer_word_level.py:97-112): -> [B, T, V]."""
    enc = tfm_encoder(sd, img_embed, enc_heads)
    emb = sd["caption_embedding.weight"][captions]                     # [B, T, E]
    return tfm_decoder(sd, emb.permute(1, 0, 2), enc, dec_heads).permute(1, 0, 2)


def tfm_greedy(sd, img_embed_row, enc_heads, dec_heads, max_len, start, end, stop_at_end=True):
    """Greedy form of generate_caption (TransformerTest.py:126-164): one image, the whole model re-run on the
    growing prefix every step, next token = argmax of the last position; stop BEFORE appending the end token.
    Returns (ids without start, logits [steps, V] of the last position)."""
    caption, logits = [start], []
    for _ in range(max_len):
        cap = torch.tensor(caption).unsqueeze(0)
        pred = tfm_forward_features(sd, img_embed_row.unsqueeze(0), cap, enc_heads, dec_heads)
        last = pred[0, -1]
        logits.append(last)
        tok = int(last.argmax())
        if stop_at_end and tok == end:
            break
        caption.append(tok)
    return caption[1:], torch.stack(logits)


# --------------------------------------------------------------------------- Bahdanau variant


def bah_encoder(sd, img_features, p="encoder"):
    """TransformerEncoderBlock.forward of transformer_word_level_bahdnau.py:70-90 on [B, 1, E]:
    permute -> (L=1, N=B); the tail replaces x by fc_out, so the result is LayerNorm2(2*fc_out), shape [1, B, E]."""
    x = img_features.unsqueeze(1).permute(1, 0, 2)                      # [1, B, E]
    a = mha(sd, p + ".attention", x, x, x, 1)                           # one key: the head count cannot matter
    x = _ln(sd, p + ".layer_norm1", x + a)
    flat = x.reshape(-1, x.shape[-1])
    h = torch.relu(_lin(sd, p + ".fc.0", flat))
    f = _lin(sd, p + ".fc.4", _bn_eval(sd, p + ".fc.3", h))
    x = f.view(x.shape[0], x.shape[1], -1)
    return _ln(sd, p + ".layer_norm2", x + f)


def bah_attention(sd, features, hidden, p="decoder.attention"):
    """Attention.forward (transformer_word_level_bahdnau.py:102-110): features [1, B, E], hidden [B, H];
    the [1,B,A] + [B,1,A] broadcast makes scores [B, B] and the softmax run over the batch's images."""
    u = _lin(sd, p + ".U", features)
    w = _lin(sd, p + ".W", hidden).unsqueeze(1)
    scores = _lin(sd, p + ".A", torch.tanh(u + w)).squeeze(2)
    alpha = torch.softmax(scores, dim=1)
    return alpha, (features * alpha.unsqueeze(2)).sum(dim=1)


def bah_decoder(sd, captions, encoder_out, p="decoder"):
    """BahdanauDecoder.forward (transformer_word_level_bahdnau.py:127-141), eval mode -> [B, T, V]."""
    B = captions.shape[0]
    H = sd[p + ".lstm.weight_hh"].shape[1]
    h, c = torch.zeros(B, H), torch.zeros(B, H)
    outs = []
    for t in range(captions.shape[1]):
        emb = sd[p + ".embedding.weight"][captions[:, t]]
        _, ctx = bah_attention(sd, encoder_out, h, p + ".attention")
        h, c = lstm_cell(sd[p + ".lstm.weight_ih"], sd[p + ".lstm.weight_hh"], sd[p + ".lstm.bias_ih"],
                         sd[p + ".lstm.bias_hh"], torch.cat((emb, ctx), dim=1), h, c)
        outs.append(_bn_eval(sd, p + ".fc.1", _lin(sd, p + ".fc.0", h)))
    return torch.stack(outs, 1)


def bah_forward_features(sd, img_features, captions):
    """ImageCaptioningModel.forward after the CNN (transformer_word_level_bahdnau.py:161-166)."""
    return bah_decoder(sd, captions, bah_encoder(sd, img_features))


def bah_greedy(sd, img_feature_row, max_len, start=2, end=3, stop_at_end=True):
    """Greedy form of generate_caption (TransformerTest_bahdnau.py:153-191), one image, prefix re-run."""
    caption, logits = [start], []
    for _ in range(max_len):
        pred = bah_forward_features(sd, img_feature_row.unsqueeze(0), torch.tensor(caption).unsqueeze(0))
        last = pred[0, -1]
        logits.append(last)
        tok = int(last.argmax())
        if stop_at_end and tok == end:
            break
        caption.append(tok)
    return caption[1:], torch.stack(logits)


# --------------------------------------------------------------------------- losses


def compute_loss(predictions, targets, padding_idx):
    """data.py:244-252 / datachar.py:234-242: masked CE averaged over ALL positions, pads in the denominator."""
    mask = (targets != padding_idx).float()
    ce = F.cross_entropy(predictions.reshape(-1, predictions.size(-1)), targets.reshape(-1), reduction="none")
    return (ce * mask.view(-1)).mean()


def seq2seq_loss(outputs, captions, pad_idx):
    """seq2seqbasic.py:204,219: CrossEntropyLoss(ignore_index=PAD), mean over non-pad positions."""
    return F.cross_entropy(outputs.reshape(-1, outputs.size(-1)), captions.reshape(-1), ignore_index=pad_idx)
