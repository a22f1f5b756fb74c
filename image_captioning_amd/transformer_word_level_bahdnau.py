"""Word-level captioner with Bahdanau additive attention and an LSTMCell decoder.

Drop-in for the model classes of the reference's transformer_word_level_bahdnau.py:53-166:
same names, constructor arguments, ``forward`` signatures and ``state_dict`` keys;
arithmetic in ``libicap_hip.so``: the inference kernels under ``model.eval()``, the training-mode
forward with a hand-written backward (``_bah_train``) under ``model.train()``.  (The reference's
``TransformerEncoderBlock_antiguo``, lines 29-50, cannot be instantiated and has
no counterpart here.)

Behaviour kept exactly as the reference computes it:

* the encoder block receives ``[B, 1, E]`` and permutes it to ``[1, B, E]``: every
  image attends over a single key (itself), so attention = out_proj(W_v x + b_v);
* its tail overwrites the residual: the block returns ``LayerNorm2(2 * fc_out)``
  with shape ``[1, B, E]`` (lines 85-90);
* ``Attention`` broadcasts ``U(features) [1, B, A]`` against ``W(h) [B, 1, A]``:
  scores are ``[B, B]`` (caption i x image j) and the softmax runs over the images
  of the batch (lines 102-110);
* logits pass through ``BatchNorm1d(vocab)`` (running statistics in eval mode).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401  (name kept for parity with the reference module)

from . import ops
from ._tfm_common import MAX_LOCKSTEP, _f32, _mha_weights, cut_before_end, pad_vocab_rows
from .backbones import EfficientNet, PreparedCache, fold_bn, get_cnn_model, require_eval  # noqa: F401


class TransformerEncoderBlock(nn.Module):
    """transformer_word_level_bahdnau.py:53-90."""

    def __init__(self, embed_dim, dense_dim, num_heads, dropout=0.3):
        super().__init__()
        self.attention = nn.MultiheadAttention(embed_dim, num_heads)
        self.layer_norm1 = nn.LayerNorm(embed_dim)
        self.layer_norm2 = nn.LayerNorm(embed_dim)
        self.fc = nn.Sequential(nn.Linear(embed_dim, dense_dim), nn.ReLU(), nn.Dropout(dropout),
                                nn.BatchNorm1d(dense_dim), nn.Linear(dense_dim, embed_dim), nn.Dropout(dropout))
        self.dropout = nn.Dropout(dropout)
        self._prep = PreparedCache()

    def _build(self):
        a = _mha_weights(self.attention)
        # eval BatchNorm1d between ReLU and the second Linear is an affine map of that Linear's input:
        # W4 (r*s + b) + b4 = (W4*s) r + (W4 b + b4)
        s, b = fold_bn(self.fc[3])
        w4, b4 = self.fc[4].weight.detach().float(), self.fc[4].bias.detach().float()
        return {"att": a,
                "ln1": (_f32(self.layer_norm1.weight), _f32(self.layer_norm1.bias)),
                "ln2": (_f32(self.layer_norm2.weight), _f32(self.layer_norm2.bias)),
                "fc0": (_f32(self.fc[0].weight), _f32(self.fc[0].bias)),
                "fc4": ((w4 * s[None, :]).contiguous(), (w4 @ b + b4).contiguous())}

    dropout_masks = None   # tests: {"enc_att", "enc_fc1", "enc_fc2"} -> factor tensors, instead of generated masks

    def rows_forward(self, rows):
        """rows [B, E] -> [B, E]; images are independent of each other in eval mode, coupled by BatchNorm1d's
        batch statistics in train mode."""
        if self.training:
            from ._bah_train import encoder_block_train
            ops.bump_weights_generation()          # BatchNorm running statistics are written by raw pointer
            return encoder_block_train(self, rows, self.dropout_masks)
        p = self._prep.get(self, self._build)
        a = p["att"]
        rows = rows.float().contiguous()
        att = ops.linear(ops.linear(rows, a["wv"], a["bv"]), a["wo"], a["bo"])
        x1 = ops.layernorm(att, *p["ln1"], res=rows)
        h = ops.linear(x1, *p["fc0"], act=ops.ACT_RELU)
        f = ops.linear(h, *p["fc4"])
        return ops.layernorm(f, *p["ln2"], res=f)            # LayerNorm2(fc_out + fc_out)

    def forward(self, x):
        """x [B, 1, E] -> [1, B, E]."""
        B, one, E = x.shape
        if one != 1:
            raise ValueError("TransformerEncoderBlock expects image features as [B, 1, E]")
        return self.rows_forward(x.reshape(B, E)).view(1, B, E)


class Attention(nn.Module):
    """Additive attention (transformer_word_level_bahdnau.py:95-110)."""

    def __init__(self, encoder_dim, decoder_dim, attention_dim):
        super().__init__()
        self.W = nn.Linear(decoder_dim, attention_dim)
        self.U = nn.Linear(encoder_dim, attention_dim)
        self.A = nn.Linear(attention_dim, 1)

    def forward(self, features, hidden_state):
        """features [1, B, E], hidden_state [B, H] -> alpha [B, B], context [B, E]."""
        feats = features.reshape(-1, features.shape[-1]).float().contiguous()
        u = ops.linear(feats, _f32(self.U.weight), _f32(self.U.bias))
        w = ops.linear(hidden_state.float().contiguous(), _f32(self.W.weight), _f32(self.W.bias))
        if u.shape[0] != w.shape[0]:
            raise ValueError("Attention: the reference's broadcast needs as many feature rows as hidden rows")
        return ops.bahdanau_attention(u, w, _f32(self.A.weight).view(-1), float(self.A.bias.item()), feats)


class BahdanauDecoder(nn.Module):
    """transformer_word_level_bahdnau.py:114-146."""

    def __init__(self, embed_dim, encoder_dim, decoder_dim, attention_dim, vocab_size, dropout=0.3):
        super().__init__()
        self.attention = Attention(encoder_dim, decoder_dim, attention_dim)
        self.embedding = nn.Embedding(vocab_size, embed_dim)
        self.lstm = nn.LSTMCell(embed_dim + encoder_dim, decoder_dim)
        self.fc = nn.Sequential(nn.Linear(decoder_dim, vocab_size), nn.BatchNorm1d(vocab_size))
        self.dropout = nn.Dropout(dropout)
        self.embed_dim, self.encoder_dim, self.decoder_dim = embed_dim, encoder_dim, decoder_dim
        self._prep = PreparedCache()

    def _build(self):
        E = self.embed_dim
        w_ih = self.lstm.weight_ih.detach().float()
        scale, bias = fold_bn(self.fc[1], self.fc[0].bias)
        att = self.attention
        fc_w4, fc_bias4, fc_scale4 = pad_vocab_rows(_f32(self.fc[0].weight), bias, scale)
        return {"emb": _f32(self.embedding.weight), "fc_w4": fc_w4, "fc_bias4": fc_bias4, "fc_scale4": fc_scale4,
                "w_ih": w_ih.contiguous(), "w_ih_e": w_ih[:, :E].contiguous(), "w_ih_c": w_ih[:, E:].contiguous(),
                "w_hh": _f32(self.lstm.weight_hh), "b_ih": _f32(self.lstm.bias_ih), "b_hh": _f32(self.lstm.bias_hh),
                "fc_w": _f32(self.fc[0].weight), "fc_scale": scale, "fc_bias": bias,
                "U": (_f32(att.U.weight), _f32(att.U.bias)), "W": (_f32(att.W.weight), _f32(att.W.bias)),
                "A_w": _f32(att.A.weight).view(-1), "A_b": float(att.A.bias.item())}

    def _weights(self):
        return self._prep.get(self, self._build)

    def init_hidden_state(self, batch_size, encoder_dim):
        dev = next(self.parameters()).device
        return (torch.zeros(batch_size, encoder_dim, device=dev), torch.zeros(batch_size, encoder_dim, device=dev))

    dropout_masks = None   # tests: {"dec_out": [B, T, H] factors}, instead of generated masks

    def forward(self, captions, encoder_out):
        """captions [B, T], encoder_out [1, B, E] -> logits [B, T, V]; attention couples the batch."""
        B, T = captions.shape
        if encoder_out.numel() != B * encoder_out.shape[-1]:
            raise ValueError("BahdanauDecoder: encoder_out must hold one row per caption")
        if self.training:
            from ._bah_train import decoder_train
            ops.bump_weights_generation()
            return decoder_train(self, captions, encoder_out.reshape(B, encoder_out.shape[-1]), self.dropout_masks)
        p = self._weights()
        feats = encoder_out.reshape(-1, encoder_out.shape[-1]).float().contiguous()
        h, c = self.init_hidden_state(B, encoder_out.size(-1))
        H = self.decoder_dim
        emb, _ = ops.embedding(p["emb"], captions.t().contiguous().view(-1))               # time-major rows
        gi_emb = ops.linear(emb, p["w_ih_e"], p["b_ih"]).view(T, B, 4 * H)                  # embedding half of W_ih, all steps
        u = ops.linear(feats, *p["U"])                                                      # step-invariant
        hs = torch.empty((B, T, H), device=feats.device, dtype=torch.float32)
        for t in range(T):
            w = ops.linear(h, *p["W"])
            _, ctx = ops.bahdanau_attention(u, w, p["A_w"], p["A_b"], feats, want_alpha=False)
            g = ops.linear(ctx, p["w_ih_c"], residual=gi_emb[t])
            g = ops.linear(h, p["w_hh"], p["b_hh"], residual=g)
            h, c = ops.lstm_pointwise(g, c)
            hs[:, t].copy_(h)
        logits = ops.linear(hs.view(B * T, H), p["fc_w"], bias=p["fc_bias"], scale=p["fc_scale"])
        return logits.view(B, T, -1)


class ImageCaptioningModel(nn.Module):
    """transformer_word_level_bahdnau.py:149-166.  <START> = 2, <END> = 3 (data.py:130)."""

    start_token, end_token = 2, 3

    def __init__(self, cnn_model, encoder, decoder):
        super().__init__()
        self.cnn_model = cnn_model
        self.encoder = encoder
        self.decoder = decoder

    def forward(self, images, captions):
        return self.forward_features(self.cnn_model(images), captions)

    def forward_features(self, img_features, captions):
        encoder_out = self.encoder(img_features.unsqueeze(1))
        return self.decoder(captions, encoder_out)

    def greedy_ids_from_features(self, img_features, max_len, want_logits=False, lockstep=None, mode=None, check=True):
        """Greedy decoding of every image as its own batch of one (generate_caption of
        TransformerTest_bahdnau.py:153-191 with argmax): the attention weight over a single image is 1, the
        context is that image's encoder row, and re-running the growing prefix equals carrying (h, c).
        mode="persistent": whole captions as single launches with LDS-resident weights (ops.greedy_bah_persist; opt-in: it needs
        the device to itself and - unless check=False - reads the ids back to detect a launch that could not run, which shows as ids of -1)."""
        require_eval(self, "ImageCaptioningModel")
        p = self.decoder._weights()
        ctx = self.encoder.rows_forward(img_features)                                        # [B, E]
        B = ctx.shape[0]
        start = torch.full((B,), self.start_token, device=ctx.device, dtype=torch.int64)
        if mode == "persistent":
            nb = ops.greedy_bah_persist_max_nb(self.decoder.decoder_dim, p["fc_w"].shape[0])
            if nb == 0:
                raise ops.IcapError("persistent decoder: shape not supported")
            ids, logits = [], []
            for s in range(0, B, nb):
                i, lg = ops.greedy_bah_persist(p, ctx[s:s + nb].contiguous(), start[s:s + nb].contiguous(), max_len, want_logits, check=check)
                ids.append(i)
                logits.append(lg)
            ids = torch.cat(ids, 0)
            return (ids, torch.cat(logits, 1)) if want_logits else ids
        if mode not in (None, "steps"):
            raise ValueError(f"unknown mode {mode!r}")
        lock = MAX_LOCKSTEP if lockstep is None else lockstep
        if B <= lock:
            ids, logits = [], []
            for s in range(0, B, MAX_LOCKSTEP):
                i, lg = ops.greedy_bah(p, ctx[s:s + MAX_LOCKSTEP].contiguous(), start[s:s + MAX_LOCKSTEP].contiguous(),
                                       max_len, want_logits)
                ids.append(i)
                logits.append(lg)
            ids = torch.cat(ids, 0)
            return (ids, torch.cat(logits, 1)) if want_logits else ids
        return self._greedy_gemm(p, ctx, start, max_len, want_logits)

    def _greedy_gemm(self, p, ctx, start, max_len, want_logits):
        B, H = ctx.shape[0], self.decoder.decoder_dim
        dev = ctx.device
        ids = torch.empty((B, max_len), device=dev, dtype=torch.int64)
        h = torch.zeros((B, H), device=dev, dtype=torch.float32)
        c = torch.zeros((B, H), device=dev, dtype=torch.float32)
        g_ctx = ops.linear(ctx, p["w_ih_c"], p["b_ih"])                                     # context half is step-invariant
        emb, _ = ops.embedding(p["emb"], start)
        all_logits = []
        for t in range(max_len):
            g = ops.linear(emb, p["w_ih_e"], residual=g_ctx)
            g = ops.linear(h, p["w_hh"], p["b_hh"], residual=g)
            h, c = ops.lstm_pointwise(g, c)
            pv, pi, logits = ops.linear_argmax(h, p["fc_w4"], bias=p["fc_bias4"], scale=p["fc_scale4"], want_logits=want_logits)
            last = t + 1 == max_len
            emb = ops.next_token(pv, pi, ids, t, None if last else p["emb"])
            if want_logits:
                all_logits.append(logits[:, :p["fc_w"].shape[0]])
        return (ids, torch.stack(all_logits, 0)) if want_logits else ids

    def caption_ids(self, images, max_len=10):
        ids = self.greedy_ids_from_features(self.cnn_model(images), max_len).tolist()
        return [cut_before_end(r, self.end_token) for r in ids]


def build_model(vocab: int, dropout: float = 0.5, weights_path=None) -> ImageCaptioningModel:
    """The configuration trained by the reference (transformer_word_level_bahdnau.py:183-188)."""
    enc = TransformerEncoderBlock(embed_dim=512, dense_dim=512, num_heads=8, dropout=dropout)
    dec = BahdanauDecoder(embed_dim=512, encoder_dim=512, decoder_dim=512, attention_dim=256, vocab_size=vocab, dropout=dropout)
    return ImageCaptioningModel(get_cnn_model(weights_path), enc, dec)
