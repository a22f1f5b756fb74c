"""Shared implementation of the word-level and char-level transformer captioners.

Both reference files (transformer_word_level.py:28-112, transformercharmiguel.py:30-113)
define the same three classes and differ only in where the vocabulary size comes
from and in head counts.  The classes here hold the parameters under the
reference's names and run their eval-mode forward through ``libicap_hip.so``.

Semantics that the kernels reproduce on purpose (they are what the reference
computes, even where they look accidental):

* the encoder block feeds ``[B, 1, E]`` to a sequence-first ``nn.MultiheadAttention``,
  so the B images of a batch attend to each other (sequence length B, batch 1);
* the decoder block tiles the B encoder rows and gives them to every batch column
  as keys: each caption position attends over all B images (tiling a key T times
  does not change a softmax-weighted sum, so the kernels use the B unique rows);
* the decoder has no self-attention in use, no causal mask and no positional
  encoding: logits at position t depend on token t and the batch's images only.
  ``attention`` / ``layer_norm1`` of the decoder are parameters that never run.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.nn import Dropout

from . import ops
from .backbones import PreparedCache, require_eval

MAX_LOCKSTEP = 8
# greedy decoding by next-token table (ImageCaptioningModel._greedy_table) when the vocabulary is at most this many times
# the caption length (one batched pass over V rows per image replaces max_len sequential passes over one row per image:
# more rows, but a single dependent phase) and the table stays small
TABLE_MAX_VOCAB_PER_STEP = 2
TABLE_MAX_ROWS = 1 << 17


def _f32(t):
    return t.detach().float().contiguous()


def pad_vocab_rows(w, bias, scale=None):
    """Pad a vocabulary projection to a multiple of 4 rows (the fused arg-max GEMM writes 16-byte groups):
    zero weight rows with a bias of -3e38 can never win the arg-max."""
    V = w.shape[0]
    pad = (-V) % 4
    if pad == 0:
        return w, bias, scale
    w = torch.cat([w, w.new_zeros(pad, w.shape[1])]).contiguous()
    bias = torch.cat([bias, bias.new_full((pad,), -3.0e38)]).contiguous()
    if scale is not None:
        scale = torch.cat([scale, scale.new_ones(pad)]).contiguous()
    return w, bias, scale


def _mha_weights(mha: nn.MultiheadAttention):
    E = mha.embed_dim
    w, b = mha.in_proj_weight, mha.in_proj_bias
    return {
        "heads": mha.num_heads,
        "wq": _f32(w[:E]), "wk": _f32(w[E:2 * E]), "wv": _f32(w[2 * E:]),
        "bq": _f32(b[:E]), "bk": _f32(b[E:2 * E]), "bv": _f32(b[2 * E:]),
        "wo": _f32(mha.out_proj.weight), "bo": _f32(mha.out_proj.bias),
    }


class TransformerEncoderBlockBase(nn.Module):
    def __init__(self, embed_dim, dense_dim, num_heads, dropout=0.3):
        super().__init__()
        self.attention = nn.MultiheadAttention(embed_dim, num_heads)
        self.layer_norm1 = nn.LayerNorm(embed_dim)
        self.layer_norm2 = nn.LayerNorm(embed_dim)
        self.fc = nn.Sequential(nn.Linear(embed_dim, dense_dim), nn.ReLU(), Dropout(dropout),
                                nn.Linear(dense_dim, embed_dim), Dropout(dropout))
        self.dropout = Dropout(dropout)
        self._prep = PreparedCache()

    def _build(self):
        return {"att": _mha_weights(self.attention),
                "ln1": (_f32(self.layer_norm1.weight), _f32(self.layer_norm1.bias)),
                "ln2": (_f32(self.layer_norm2.weight), _f32(self.layer_norm2.bias)),
                "fc0": (_f32(self.fc[0].weight), _f32(self.fc[0].bias)),
                "fc3": (_f32(self.fc[3].weight), _f32(self.fc[3].bias))}

    def _ffn_tail(self, p, x, attn):
        x1 = ops.layernorm(attn, *p["ln1"], res=x)
        h = ops.linear(x1, *p["fc0"], act=ops.ACT_RELU)
        f = ops.linear(h, *p["fc3"])
        return ops.layernorm(f, *p["ln2"], res=x1)

    def forward(self, x):
        """x [1, B, E] -> [B, 1, E]; self-attention runs ACROSS the B images (transformer_word_level.py:43-49)."""
        require_eval(self, "TransformerEncoderBlock")
        p = self._prep.get(self, self._build)
        a = p["att"]
        one, B, E = x.shape
        if one != 1:
            raise ValueError("TransformerEncoderBlock expects the image embedding as [1, B, E]")
        rows = x.reshape(B, E).float().contiguous()
        q = ops.linear(rows, a["wq"], a["bq"])
        k = ops.linear(rows, a["wk"], a["bk"])
        v = ops.linear(rows, a["wv"], a["bv"])
        att = ops.mha_core(q.view(B, 1, E), k.view(B, 1, E), v.view(B, 1, E), a["heads"]).view(B, E)
        att = ops.linear(att, a["wo"], a["bo"])
        return self._ffn_tail(p, rows, att).view(B, 1, E)

    def forward_independent(self, rows):
        """Each row as its own batch of one (what generate_caption computes image by image): with a single
        key the attention output is out_proj(W_v x + b_v).  rows [B, E] -> [B, E]."""
        require_eval(self, "TransformerEncoderBlock")
        p = self._prep.get(self, self._build)
        a = p["att"]
        rows = rows.float().contiguous()
        att = ops.linear(ops.linear(rows, a["wv"], a["bv"]), a["wo"], a["bo"])
        return self._ffn_tail(p, rows, att)


class TransformerDecoderBlockBase(nn.Module):
    def __init__(self, embed_dim, ff_dim, num_heads, vocab_size, dropout=0.3):
        super().__init__()
        self.attention = nn.MultiheadAttention(embed_dim, num_heads)        # never used by forward (as in the reference)
        self.cross_attention = nn.MultiheadAttention(embed_dim, num_heads)
        self.layer_norm1 = nn.LayerNorm(embed_dim)                           # never used by forward
        self.layer_norm2 = nn.LayerNorm(embed_dim)
        self.layer_norm3 = nn.LayerNorm(embed_dim)
        self.fc = nn.Sequential(nn.Linear(embed_dim, ff_dim), nn.ReLU(), Dropout(dropout),
                                nn.Linear(ff_dim, embed_dim), Dropout(dropout))
        self.output_fc = nn.Linear(embed_dim, vocab_size)
        self.dropout = Dropout(dropout)
        self._prep = PreparedCache()

    def _build(self):
        return {"att": _mha_weights(self.cross_attention),
                "ln2": (_f32(self.layer_norm2.weight), _f32(self.layer_norm2.bias)),
                "ln3": (_f32(self.layer_norm3.weight), _f32(self.layer_norm3.bias)),
                "fc0": (_f32(self.fc[0].weight), _f32(self.fc[0].bias)),
                "fc3": (_f32(self.fc[3].weight), _f32(self.fc[3].bias)),
                "out": (_f32(self.output_fc.weight), _f32(self.output_fc.bias))}

    def _rows_forward(self, p, x_rows, enc_rows):
        """x_rows [R, E] caption embeddings (any row order), enc_rows [B, E] -> logits [R, V]."""
        a = p["att"]
        R, E = x_rows.shape
        Bk = enc_rows.shape[0]
        q = ops.linear(x_rows, a["wq"], a["bq"])
        k = ops.linear(enc_rows, a["wk"], a["bk"])
        v = ops.linear(enc_rows, a["wv"], a["bv"])
        ca = ops.mha_core(q.view(R, 1, E), k.view(Bk, 1, E), v.view(Bk, 1, E), a["heads"]).view(R, E)
        ca = ops.linear(ca, a["wo"], a["bo"])
        x = ops.layernorm(ca, *p["ln2"], res=x_rows)
        h = ops.linear(x, *p["fc0"], act=ops.ACT_RELU)
        f = ops.linear(h, *p["fc3"])
        x = ops.layernorm(f, *p["ln3"], res=x)
        return ops.linear(x, *p["out"])

    def forward(self, x, encoder_out):
        """x [T, B, E], encoder_out [B, 1, E] -> logits [T, B, V]  (transformer_word_level.py:72-79)."""
        require_eval(self, "TransformerDecoderBlock")
        p = self._prep.get(self, self._build)
        T, B, E = x.shape
        logits = self._rows_forward(p, x.reshape(T * B, E).float().contiguous(),
                                    encoder_out.reshape(-1, E).float().contiguous())
        return logits.view(T, B, -1)

    def cross_attention_single(self, enc_rows):
        """Cross-attention output when the batch holds one image: out_proj(W_v enc + b_v), [B, E]."""
        p = self._prep.get(self, self._build)
        a = p["att"]
        return ops.linear(ops.linear(enc_rows.float().contiguous(), a["wv"], a["bv"]), a["wo"], a["bo"])


class ImageCaptioningModelBase(nn.Module):
    start_token = 2
    end_token = 3

    def __init__(self, cnn_model, encoder, decoder, vocab_size, num_captions_per_image=5, embed_dim=512):
        super().__init__()
        self.cnn_model = cnn_model
        self.encoder = encoder
        self.decoder = decoder
        self.num_captions_per_image = num_captions_per_image
        self.caption_embedding = nn.Embedding(vocab_size, embed_dim)
        self._prep = PreparedCache()

    dropout_masks = None   # tests: factor tensors instead of generated dropout masks (see _tfm_train.model_train_forward)

    def forward(self, images, captions):
        """images [B,3,H,W], captions [B,T] int64 -> logits [B,T,V]  (transformer_word_level.py:92-112)."""
        img_embed = self.cnn_model(images)                                  # [B, E]
        return self.forward_features(img_embed, captions)

    def forward_features(self, img_embed, captions):
        """Same as forward but from the CNN output [B, E]."""
        if self.training:
            from ._tfm_train import model_train_forward
            ops.bump_weights_generation()
            return model_train_forward(self, img_embed, captions, self.dropout_masks)
        B, T = captions.shape
        E = img_embed.shape[-1]
        enc = self.encoder(img_embed.view(1, B, E))                         # [B, 1, E], coupled across the batch
        emb, err = ops.embedding(_f32(self.caption_embedding.weight), captions.contiguous().view(-1))  # rows (b, t)
        p = self.decoder._prep.get(self.decoder, self.decoder._build)
        logits = self.decoder._rows_forward(p, emb, enc.view(B, E))
        return logits.view(B, T, -1)

    # -- greedy captioning ---------------------------------------------------------------------
    def _greedy_weights(self):
        def build():
            p = self.decoder._build()
            return {"emb": _f32(self.caption_embedding.weight),
                    "ln2_g": p["ln2"][0], "ln2_b": p["ln2"][1], "ln3_g": p["ln3"][0], "ln3_b": p["ln3"][1],
                    "fc1_w": p["fc0"][0], "fc1_b": p["fc0"][1], "fc2_w": p["fc3"][0], "fc2_b": p["fc3"][1],
                    "out_w": p["out"][0], "out_b": p["out"][1]}
        def build_all():
            wd = build()
            wd["out_w4"], wd["out_b4"], _ = pad_vocab_rows(wd["out_w"], wd["out_b"])
            return wd
        return self._prep.get(self, build_all)

    def greedy_ids_from_features(self, img_embed, max_len, want_logits=False, lockstep=None, mode=None):
        """Greedy decoding of every image as its own batch of one (the semantics of generate_caption,
        TransformerTest.py:126-164, with argmax in place of top-k sampling).  The CNN / encoder / cross-attention
        results do not depend on the step, so they are computed once.  Returns ids [B, max_len] on the device.
        mode None picks by shape: "table" (vocabulary small against the caption length - the char model: one batched
        decoder pass + look-ups), "steps" (<= lockstep captions, 2-3 launches per step), "gemm" (large batches on the
        matrix cores); passing lockstep chooses between the last two only; "persistent" asks for the single-launch
        decoder with LDS-resident weights."""
        require_eval(self, "ImageCaptioningModel")
        wd = self._greedy_weights()
        enc = self.encoder.forward_independent(img_embed)
        ca = self.decoder.cross_attention_single(enc)                        # [B, E]
        B = ca.shape[0]
        start = torch.full((B,), self.start_token, device=ca.device, dtype=torch.int64)
        lock = MAX_LOCKSTEP if lockstep is None else lockstep
        V = wd["out_w"].shape[0]
        if mode is None:
            table = lockstep is None and V <= TABLE_MAX_VOCAB_PER_STEP * max_len and V * B <= TABLE_MAX_ROWS
            mode = "table" if table else ("steps" if B <= lock else "gemm")
        if mode == "table":
            return self._greedy_table(wd, ca, start, max_len, want_logits)
        if mode == "persistent":
            return self._greedy_persistent(wd, ca, start, max_len, want_logits)
        if mode == "steps" or (mode != "gemm" and B <= lock):
            return self._greedy_small(wd, ca, start, max_len, want_logits)
        return self._greedy_gemm(wd, ca, start, max_len, want_logits)

    def _greedy_table(self, wd, ca, start, max_len, want_logits):
        """Small vocabularies (the 81-symbol char model).  The decoder never looks at earlier positions (no self-attention
        or positional encoding is called, transformercharmiguel.py:73-80), so the logits after token v for image b are a
        function of (v, b) alone: ONE batched decoder pass over the V x B rows LN2(emb[v] + ca[b]) gives the table
        next[v][b], and the caption is max_len look-ups (icap_table_walk) - result-identical to running the steps one
        after the other, without their max_len sequential launches."""
        B, E = ca.shape
        V = wd["out_w"].shape[0]
        tok = torch.arange(V, device=ca.device, dtype=torch.int64).repeat_interleave(B)     # one small launch; nothing is kept per batch size
        rows, _ = ops.embedding(wd["emb"], tok)                                       # row v*B + b = emb[v]
        x = ops.layernorm(rows, wd["ln2_g"], wd["ln2_b"], res=ca)                     # residual row (v*B + b) % B = ca[b]
        h = ops.linear(x, wd["fc1_w"], wd["fc1_b"], act=ops.ACT_RELU)
        f = ops.linear(h, wd["fc2_w"], wd["fc2_b"], residual=x)
        x2 = ops.layernorm(f, wd["ln3_g"], wd["ln3_b"])
        pv, pi, logits = ops.linear_argmax(x2, wd["out_w4"], wd["out_b4"], want_logits=want_logits)
        nxt = torch.empty((V * B, 1), device=ca.device, dtype=torch.int64)
        ops.next_token(pv, pi, nxt, 0)
        ids, lg = ops.table_walk(nxt.view(V, B), start, max_len, logits_tab=logits if want_logits else None, V=V)
        return (ids, lg) if want_logits else ids

    def _greedy_persistent(self, wd, ca, start, max_len, want_logits):
        """One persistent launch per group of captions with the step's weights resident in LDS
        (icap_greedy_tfm_persist_f32); measured slower than the table walk for the char model and level with the
        step-per-launch driver for the word model, so it is never chosen by default."""
        E, FF, V = ca.shape[1], wd["fc1_w"].shape[0], wd["out_w"].shape[0]
        nb = ops.greedy_tfm_persist_max_nb(E, FF, V)
        if nb == 0:
            raise ops.IcapError("persistent decoder: shape not supported")
        ids, logits = [], []
        for s in range(0, ca.shape[0], nb):
            i, lg = ops.greedy_tfm_persist(wd, ca[s:s + nb].contiguous(), start[s:s + nb].contiguous(), max_len, want_logits)
            ids.append(i)
            logits.append(lg)
        ids = torch.cat(ids, 0)
        return (ids, torch.cat(logits, 1)) if want_logits else ids

    def _greedy_small(self, wd, ca, start, max_len, want_logits):
        ids, logits = [], []
        for s in range(0, ca.shape[0], MAX_LOCKSTEP):
            i, lg = ops.greedy_tfm(wd, ca[s:s + MAX_LOCKSTEP].contiguous(), start[s:s + MAX_LOCKSTEP].contiguous(),
                                   max_len, want_logits)
            ids.append(i)
            logits.append(lg)
        ids = torch.cat(ids, 0)
        return (ids, torch.cat(logits, 1)) if want_logits else ids

    def _greedy_gemm(self, wd, ca, start, max_len, want_logits):
        """Large image batches: each step is three [B, *] GEMMs on the matrix cores plus two row kernels; the
        vocabulary GEMM emits arg-max partials in its epilogue, and one kernel turns them into the token id and
        the next step's LayerNorm2(embedding + cross-attention) row, so logits never make a round trip."""
        B = ca.shape[0]
        ids = torch.empty((B, max_len), device=ca.device, dtype=torch.int64)
        xin, _ = ops.embedding(wd["emb"], start, add=ca)
        x = ops.layernorm(xin, wd["ln2_g"], wd["ln2_b"])
        all_logits = []
        bf16 = ops.compute_dtype() == "bf16" and wd["fc1_w"].shape[1] % 8 == 0
        if bf16 and "fc1_w16" not in wd:
            wd["fc1_w16"], wd["out_w16"] = ops.to_bf16(wd["fc1_w"]), ops.to_bf16(wd["out_w4"])
        V = wd["out_w"].shape[0]
        x3 = ops.compute_dtype() == "f32x3" and wd["fc1_w"].shape[1] % 8 == 0 and B >= 256
        if x3 and "out_w3" not in wd:
            wd["out_w3"] = ops.split3(wd["out_w4"])
        for t in range(max_len):
            last = t + 1 == max_len
            if bf16:
                # bf16 products for the two wide GEMMs (fc2 has too few output tiles to gain)
                h = ops.linear_bf16(x, wd["fc1_w16"], wd["fc1_b"], act=ops.ACT_RELU)
                f = ops.linear(h, wd["fc2_w"], wd["fc2_b"], residual=x)
                x2 = ops.layernorm(f, wd["ln3_g"], wd["ln3_b"])
                pv, pi, logits = ops.linear_bf16_argmax(x2, wd["out_w16"], wd["out_b4"], want_logits=want_logits)
                x = ops.next_token(pv, pi, ids, t, None if last else wd["emb"], None if last else ca,
                                   None if last else wd["ln2_g"], None if last else wd["ln2_b"])
            elif x3:
                # fp32-accurate products on the bf16 matrix cores for the GEMMs they speed up (ops.linear_x3_auto decides)
                h = ops.linear_x3_auto(x, wd["fc1_w"], wd, "fc1_w3", wd["fc1_b"], act=ops.ACT_RELU)
                f = ops.linear_x3_auto(h, wd["fc2_w"], wd, "fc2_w3", wd["fc2_b"], residual=x)
                # 53 column tiles would each repeat the split of the LayerNorm's result: the LayerNorm kernel writes it split
                x23 = ops.layernorm(f, wd["ln3_g"], wd["ln3_b"], split=True)
                pv, pi, logits = ops.linear_f32x3_argmax(x23, wd["out_w3"], wd["out_b4"], want_logits=want_logits)
                x = ops.next_token(pv, pi, ids, t, None if last else wd["emb"], None if last else ca,
                                   None if last else wd["ln2_g"], None if last else wd["ln2_b"])
            else:
                h = ops.linear(x, wd["fc1_w"], wd["fc1_b"], act=ops.ACT_RELU)
                f = ops.linear(h, wd["fc2_w"], wd["fc2_b"], residual=x)
                x2 = ops.layernorm(f, wd["ln3_g"], wd["ln3_b"])
                pv, pi, logits = ops.linear_argmax(x2, wd["out_w4"], wd["out_b4"], want_logits=want_logits)
                x = ops.next_token(pv, pi, ids, t, None if last else wd["emb"], None if last else ca,
                                   None if last else wd["ln2_g"], None if last else wd["ln2_b"])
            if want_logits:
                all_logits.append(logits[:, :V])
        return (ids, torch.stack(all_logits, 0)) if want_logits else ids

    def sample_ids_from_features(self, img_embed, max_len, top_k=10, seed=0):
        """The reference's actual generation mode (TransformerTest.py:139-164): softmax -> top-k -> one multinomial draw per
        step, every image its own batch of one, all images in lock-step on the GPU.  Same step structure as the greedy
        GEMM path; the vocabulary projection writes full logits and the sampling head picks the token (counter-based
        generator: step t, image r uses element t*B + r of stream `seed`).  top_k = 1 reproduces greedy decoding."""
        require_eval(self, "ImageCaptioningModel")
        wd = self._greedy_weights()
        enc = self.encoder.forward_independent(img_embed)
        ca = self.decoder.cross_attention_single(enc)                        # [B, E]
        B = ca.shape[0]
        V = wd["out_w"].shape[0]
        start = torch.full((B,), self.start_token, device=ca.device, dtype=torch.int64)
        ids = torch.empty((B, max_len), device=ca.device, dtype=torch.int64)
        xin, _ = ops.embedding(wd["emb"], start, add=ca)
        x = ops.layernorm(xin, wd["ln2_g"], wd["ln2_b"])
        for t in range(max_len):
            h = ops.linear(x, wd["fc1_w"], wd["fc1_b"], act=ops.ACT_RELU)
            f = ops.linear(h, wd["fc2_w"], wd["fc2_b"], residual=x)
            x2 = ops.layernorm(f, wd["ln3_g"], wd["ln3_b"])
            logits = ops.linear(x2, wd["out_w4"], wd["out_b4"])              # vocabulary padded to a multiple of 4 columns
            ops.topk_sample(logits[:, :V], top_k, seed, t * B, ids_out=ids[:, t], want_topk=False)
            if t + 1 < max_len:
                xin, _ = ops.embedding(wd["emb"], ids[:, t].contiguous(), add=ca)
                x = ops.layernorm(xin, wd["ln2_g"], wd["ln2_b"])
        return ids

    def caption_ids(self, images, max_len=10):
        """Greedy captions (lists of ids, start token dropped, cut before the end token)."""
        feats = self.cnn_model(images)
        ids = self.greedy_ids_from_features(feats, max_len).tolist()
        return [cut_before_end(r, self.end_token) for r in ids]


def cut_before_end(row, end_token):
    out = []
    for t in row:
        if t == end_token:
            break
        out.append(int(t))
    return out
