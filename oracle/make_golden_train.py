"""Generate tests/golden/bahdanau_train_small.npz by running the REFERENCE's classes in train() mode on CPU.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_train.py       (build container only: needs /root/reference)

One teacher-forced training step of transformer_word_level_bahdnau.py:218-233 on post-CNN features with
dropout = 0 (the reference is only deterministic without dropout): forward, data.compute_loss on
predictions[:, :-1] vs captions[:, 1:], backward, one torch.optim.AdamW(lr=1e-4, weight_decay=1e-5) step.
Stored: the loss, every parameter's gradient, every parameter after the step and the BatchNorm running statistics.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden import OUT, REF, FeaturePassThrough, captions, features, install_stubs  # noqa: E402


def main():
    install_stubs()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import data as ref_data
    import transformer_word_level_bahdnau as ref_bah
    from image_captioning_amd import transformer_word_level_bahdnau as my_bah
    from image_captioning_amd.synthetic import seeded_state_dict

    E, ENC, H, A, V, DENSE, B, T, PAD = 32, 64, 64, 16, 97, 48, 6, 8, 0
    mine = my_bah.ImageCaptioningModel(FeaturePassThrough(), my_bah.TransformerEncoderBlock(ENC, DENSE, 8, dropout=0.0),
                                       my_bah.BahdanauDecoder(E, ENC, H, A, V, dropout=0.0))
    sd = seeded_state_dict(mine, seed=31)
    ref = ref_bah.ImageCaptioningModel(FeaturePassThrough(), ref_bah.TransformerEncoderBlock(ENC, DENSE, 8, dropout=0.0),
                                       ref_bah.BahdanauDecoder(E, ENC, H, A, V, dropout=0.0))
    ref.load_state_dict(sd, strict=True)
    ref.train()
    feats = features(B, ENC, 501).requires_grad_(True)
    caps = captions(B, T, V, 502)
    caps[1, 5:] = PAD
    caps[4, 3:] = PAD
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=1e-5)
    opt.zero_grad()
    pred = ref(feats, caps)
    loss = ref_data.compute_loss(pred[:, :-1], caps[:, 1:], PAD)
    loss.backward()
    res = {"dims": np.array([E, ENC, H, A, V, DENSE, B, T, PAD]), "caps": caps.numpy(), "loss": float(loss), "logits": pred.detach().numpy(),
           "grad_feats": feats.grad.numpy()}
    for k, p in ref.named_parameters():
        res["grad." + k] = p.grad.numpy() if p.grad is not None else np.zeros(0, dtype=np.float32)
    opt.step()
    for k, v in ref.state_dict().items():
        res["after." + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "bahdanau_train_small.npz"), **res)
    print("loss", float(loss), "->", OUT)

    # ---- seq2seq decoders in train() mode: forward, CrossEntropyLoss(ignore_index=PAD) as at seq2seqbasic.py:204,219, backward
    import seq2seqbasic as ref_s2s
    from image_captioning_amd import seq2seqbasic as my_s2s
    for kind in ("lstm", "gru"):
        E, H, V, L, B, T = 32, 64, 97, 2, 5, 7
        mine = my_s2s.DecoderRNN(E, H, V, num_layers=L, rnn_type=kind)
        sd = seeded_state_dict(mine, seed=11)
        ref = ref_s2s.DecoderRNN(E, H, V, num_layers=L, rnn_type=kind)
        ref.load_state_dict(sd, strict=True)
        ref.train()
        feats = features(B, E, 601).requires_grad_(True)
        caps = captions(B, T, V, 602)
        caps[0, 4:] = 0
        caps[3, 2:] = 0
        out = ref(feats, caps)
        loss = torch.nn.CrossEntropyLoss(ignore_index=0)(out.view(-1, V), caps.view(-1))
        loss.backward()
        res = {"dims": np.array([E, H, V, L, B, T]), "caps": caps.numpy(), "loss": float(loss), "logits": out.detach().numpy(),
               "grad_feats": feats.grad.numpy()}
        for k, p in ref.named_parameters():
            res["grad." + k] = p.grad.numpy()
        np.savez_compressed(os.path.join(OUT, f"seq2seq_{kind}_train_small.npz"), **res)
        print("seq2seq", kind, "train loss", float(loss))

    # ---- word / char transformer in train() mode with dropout 0: forward, data.compute_loss, backward
    import transformer_word_level as ref_twl
    import transformercharmiguel as ref_chr
    import datachar as ref_dc
    from image_captioning_amd import transformer_word_level as my_twl
    from image_captioning_amd import transformercharmiguel as my_chr
    for name, ref_mod, my_mod, (E, FF, eh, dh, V), pad in (("word", ref_twl, my_twl, (64, 128, 4, 4, 97), 0),
                                                            ("char", ref_chr, my_chr, (64, 96, 2, 4, 81), 2)):
        if name == "word":
            ref_mod.vocab_size = V
            my_mod.vocab_size = V
        else:
            assert ref_dc.NUM_CHAR == V
        B, T = 5, 6
        mine = my_mod.ImageCaptioningModel(FeaturePassThrough(), my_mod.TransformerEncoderBlock(E, FF, eh, dropout=0.0),
                                           my_mod.TransformerDecoderBlock(E, FF, dh, dropout=0.0), embed_dim=E)
        sd = seeded_state_dict(mine, seed=41)
        ref = ref_mod.ImageCaptioningModel(FeaturePassThrough(), ref_mod.TransformerEncoderBlock(E, FF, eh, dropout=0.0),
                                           ref_mod.TransformerDecoderBlock(E, FF, dh, dropout=0.0), embed_dim=E)
        ref.load_state_dict(sd, strict=True)
        ref.train()
        feats = features(B, E, 701).requires_grad_(True)
        caps = captions(B, T, V, 702, start=2 if name == "word" else 0)
        caps[1, 4:] = pad
        pred = ref(feats, caps)
        loss = ref_data.compute_loss(pred[:, :-1], caps[:, 1:], pad)
        loss.backward()
        res = {"dims": np.array([E, FF, eh, dh, V, B, T, pad]), "caps": caps.numpy(), "loss": float(loss), "logits": pred.detach().numpy(),
               "grad_feats": feats.grad.numpy()}
        for k, p in ref.named_parameters():
            res["grad." + k] = p.grad.numpy() if p.grad is not None else np.zeros(0, dtype=np.float32)
        np.savez_compressed(os.path.join(OUT, f"{name}_tfm_train_small.npz"), **res)
        print(name, "transformer train loss", float(loss))


if __name__ == "__main__":
    main()
