"""CNN encoder + LSTM/GRU decoder captioner on the MI355X kernels.

Same class names, constructor arguments, ``forward`` / ``sample`` signatures and
``state_dict`` keys as the reference's seq2seqbasic.py:15-136, so code written
against those classes runs unchanged; the arithmetic runs in ``libicap_hip.so``.

``model.eval()``: forward / sample / caption_ids run the inference kernels.  ``model.train()``: forward runs the
train-mode kernels with a hand-written backward (``_s2s_train.py``); the loop body of seq2seqbasic.py:216-223 is
``training.Trainer.step(..., loss="cross_entropy")``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import backbones, ops
from .backbones import PreparedCache, fold_bn, require_eval

MAX_LOCKSTEP = 8  # captions decoded together by one greedy-driver call


class EncoderCNN(nn.Module):
    """ResNet-18 or VGG-16 trunk -> Linear(in, embed) -> BatchNorm1d  (seq2seqbasic.py:15-45)."""

    def __init__(self, embed_size, cnn_type="resnet18", weights_path=None):
        super().__init__()
        if cnn_type.startswith("resnet"):
            if cnn_type != "resnet18":
                raise KeyError(cnn_type)
            backbone = backbones.resnet18(pretrained=True, weights_path=weights_path)
            self.cnn = backbones.ResNetTrunk(backbone)
            in_features = backbone.fc.in_features
        elif cnn_type.startswith("vgg"):
            if cnn_type != "vgg16":
                raise KeyError(cnn_type)
            backbone = backbones.vgg16(pretrained=True, weights_path=weights_path)
            self.cnn = backbones.VGGTrunk(backbone)
            in_features = 512 * 7 * 7
        else:
            raise ValueError(f"unknown cnn_type {cnn_type!r}")
        self.cnn_type = cnn_type
        self.fc = nn.Linear(in_features, embed_size)
        self.bn = nn.BatchNorm1d(embed_size, momentum=0.01)
        self._prep = PreparedCache()

    def _build(self):
        w = self.fc.weight
        if self.cnn_type.startswith("vgg"):
            w = backbones.vgg_fc_weight_nhwc(w)
        scale, bias = fold_bn(self.bn, self.fc.bias)
        return w.float().contiguous(), scale, bias

    def forward(self, images):
        if self.training:
            from ._s2s_train import EncoderHeadTrain, resnet_trunk_train
            ops.bump_weights_generation()          # BatchNorm running statistics are written by raw pointer
            vgg = self.cnn_type.startswith("vgg")
            with torch.no_grad():                                       # seq2seqbasic.py:41
                feats = self.cnn(images) if vgg else resnet_trunk_train(self.cnn, images)
            feats = feats.reshape(images.size(0), -1).contiguous()
            return EncoderHeadTrain.apply(feats, self.fc.weight, self.fc.bias, self.bn.weight, self.bn.bias, self.bn, vgg)
        feats = self.cnn(images)
        feats = feats.reshape(images.size(0), -1)
        w, scale, bias = self._prep.get(self, self._build)
        return ops.linear(feats, w, bias=bias, scale=scale)


class DecoderRNN(nn.Module):
    """Embedding -> feature_fc -> stacked LSTM/GRU -> fc  (seq2seqbasic.py:48-91)."""

    def __init__(self, embed_size, hidden_size, vocab_size, num_layers=1, rnn_type="gru"):
        super().__init__()
        self.embed = nn.Embedding(vocab_size, embed_size)
        self.feature_fc = nn.Linear(embed_size, hidden_size)
        if rnn_type == "gru":
            self.rnn = nn.GRU(hidden_size, hidden_size, num_layers, batch_first=True)
        elif rnn_type == "lstm":
            self.rnn = nn.LSTM(hidden_size, hidden_size, num_layers, batch_first=True)
        else:
            raise ValueError(f"unknown rnn_type {rnn_type!r}")
        self.fc = nn.Linear(hidden_size, vocab_size)
        self.rnn_type, self.num_layers, self.hidden_size = rnn_type, num_layers, hidden_size
        self._prep = PreparedCache()

    # -- kernel-layout weights -------------------------------------------------
    def _build(self):
        f32 = lambda t: t.detach().float().contiguous()
        L = self.num_layers
        wd = {
            "is_lstm": self.rnn_type == "lstm",
            "w_ih": [f32(getattr(self.rnn, f"weight_ih_l{l}")) for l in range(L)],
            "w_hh": [f32(getattr(self.rnn, f"weight_hh_l{l}")) for l in range(L)],
            "b_ih": [f32(getattr(self.rnn, f"bias_ih_l{l}")) for l in range(L)],
            "b_hh": [f32(getattr(self.rnn, f"bias_hh_l{l}")) for l in range(L)],
            "fc_w": f32(self.fc.weight), "fc_b": f32(self.fc.bias),
            "ff_w": f32(self.feature_fc.weight), "ff_b": f32(self.feature_fc.bias),
            "embed": f32(self.embed.weight),
        }
        # feature_fc(embed(token)) depends on the token only: one [V, H] table turns the
        # per-step embedding + Linear (seq2seqbasic.py:88-89) into a row gather.
        wd["in_table"] = ops.linear(wd["embed"], wd["ff_w"], bias=wd["ff_b"])
        return wd

    def _weights(self):
        return self._prep.get(self, self._build)

    # -- teacher-forced forward ---------------------------------------------------
    def forward(self, features, captions):
        """features [B, embed], captions [B, T] -> logits [B, T, V]  (seq2seqbasic.py:59-66)."""
        if self.training:
            from ._s2s_train import decoder_train
            ops.bump_weights_generation()
            return decoder_train(self, features, captions)
        wd = self._weights()
        B, T = captions.shape
        H = self.hidden_size
        tok = captions[:, :-1].contiguous()
        # step 0 input is the image feature, steps 1.. are embeddings of captions[:, :-1]
        x0 = ops.linear(features.float().contiguous(), wd["ff_w"], bias=wd["ff_b"])           # [B, H]
        seq = [x0]
        if T > 1:
            emb, _ = ops.embedding(wd["in_table"], tok.t().contiguous().view(-1))             # [(T-1)*B, H], time-major
            seq += list(emb.view(T - 1, B, H).unbind(0))
        layer_in = [s.contiguous() for s in seq]
        dev = features.device
        for l in range(self.num_layers):
            h = torch.zeros((B, H), device=dev, dtype=torch.float32)
            c = torch.zeros((B, H), device=dev, dtype=torch.float32)
            outs = []
            xs = torch.stack(layer_in, 0).view(T * B, -1)
            gi_all = ops.linear(xs, wd["w_ih"][l], bias=wd["b_ih"][l]).view(T, B, -1)        # input half, all steps at once
            for t in range(T):
                if wd["is_lstm"]:
                    gates = ops.linear(h, wd["w_hh"][l], bias=wd["b_hh"][l], residual=gi_all[t])
                    h, c = ops.lstm_pointwise(gates, c)
                else:
                    gh = ops.linear(h, wd["w_hh"][l], bias=wd["b_hh"][l])
                    h = ops.gru_pointwise(gi_all[t], gh, h)
                outs.append(h)
            layer_in = outs
        hs = torch.stack(layer_in, 1).view(B * T, H)                                          # batch-major rows
        logits = ops.linear(hs, wd["fc_w"], bias=wd["fc_b"])
        return logits.view(B, T, -1)

    # -- greedy decoding ------------------------------------------------------------
    def sample_ids(self, features, max_len=20, want_logits=False, mode="auto", check=True):
        """Greedy token ids for a batch of feature rows, [B, max_len] on the device (no host sync).
        Every row is decoded independently, exactly as the reference decodes one image at a time.
        mode="persistent" (two layers): whole captions as single launches with LDS-resident weights, one or two captions per
        launch (ops.greedy_rnn2_persist; opt-in: it needs the device to itself and synchronises with the host once per launch
        unless check=False)."""
        require_eval(self, "DecoderRNN")
        wd = self._weights()
        feats = features.float().contiguous().view(-1, features.shape[-1])
        x0 = ops.linear(feats, wd["ff_w"], bias=wd["ff_b"])
        ids, logits = [], []
        if mode == "persistent":
            nb = ops.greedy_rnn2_persist_max_nb(wd["is_lstm"], wd["w_hh"][0].shape[1], wd["fc_w"].shape[0]) if len(wd["w_ih"]) == 2 else 0
            if nb <= 0:
                raise ops.IcapError("DecoderRNN.sample_ids: the persistent form needs two layers whose weights fit the chip's LDS")
            step = nb
            run = lambda x: ops.greedy_rnn2_persist(wd, x, max_len, want_logits, check=check)
        elif mode == "auto":
            step = MAX_LOCKSTEP
            run = lambda x: ops.greedy_rnn(wd, x, max_len, want_logits)
        else:
            raise ValueError(f"DecoderRNN.sample_ids: unknown mode {mode!r}")
        for s in range(0, x0.shape[0], step):
            i, lg = run(x0[s:s + step].contiguous())
            ids.append(i)
            logits.append(lg)
        ids = torch.cat(ids, 0)
        return (ids, torch.cat(logits, 1)) if want_logits else ids

    def sample(self, features, max_len=20, start_idx=2, end_idx=3):
        """features [embed] -> list of token ids, cut after the first end token (seq2seqbasic.py:68-91).
        As in the reference the start token is never fed: the image feature is the first RNN input."""
        ids = self.sample_ids(features.view(1, -1), max_len)[0].tolist()
        out = []
        for t in ids:
            out.append(int(t))
            if t == end_idx:
                break
        return out


class ImageCaptioningModel(nn.Module):
    """EncoderCNN + DecoderRNN  (seq2seqbasic.py:94-103)."""

    def __init__(self, embed_size, hidden_size, vocab_size, cnn_type="resnet18", rnn_type="gru", num_layers=1):
        super().__init__()
        self.encoder = EncoderCNN(embed_size, cnn_type=cnn_type)
        self.decoder = DecoderRNN(embed_size, hidden_size, vocab_size, num_layers=num_layers, rnn_type=rnn_type)

    def forward(self, images, captions):
        return self.decoder(self.encoder(images), captions)

    def caption_ids(self, images, max_len=20, end_idx=3):
        """Greedy captions for a batch of images: encoder once, then DecoderRNN.sample per image
        (the loop of seq2seqbasic.py:152-155) without a host round trip per token."""
        feats = self.encoder(images)
        ids = self.decoder.sample_ids(feats, max_len).tolist()
        out = []
        for row in ids:
            cut = []
            for t in row:
                cut.append(int(t))
                if t == end_idx:
                    break
            out.append(cut)
        return out
