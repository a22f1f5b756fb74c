"""Generate tests/golden/ref_ckpt_*.pth + ref_ckpt_io.npz: checkpoints WRITTEN BY THE REFERENCE'S OWN CLASSES.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_ckpt.py

The reference saves a bare ``state_dict`` (``torch.save(model.state_dict(), path)``: seq2seqbasic.py:264,
transformer_word_level.py:246, transformer_word_level_bahdnau.py:298, transformercharmiguel.py:248).  Here small
instances of the reference's classes are constructed with torch's own default initialisation under a fixed seed (this
repo's code plays no part in producing the weights), their BatchNorm running statistics are made non-trivial, the
``state_dict`` is saved exactly as the reference does, and the reference's eval-mode outputs on seeded inputs are stored
next to it.  tests/test_gpu_checkpoint.py loads the files with ``checkpoint.load_checkpoint(strict=True)`` and must
reproduce the stored outputs on the GPU.  The CNN backbones come from packages that are absent here, so the reference
models are built with an identity in place of the CNN and are fed post-CNN features (as in make_golden.py).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden import (OUT, REF, FeaturePassThrough, captions, features, gaps, greedy_with_reference,  # noqa: E402
                                install_stubs)


def randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.3)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 1.5 + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            m.num_batches_tracked.fill_(17)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import seq2seqbasic as ref_s2s
    import transformer_word_level as ref_twl
    import transformer_word_level_bahdnau as ref_bah
    import transformercharmiguel as ref_chr
    torch.set_grad_enabled(False)
    io = {}

    # seq2seq decoder (LSTM, 2 layers) - seq2seqbasic.py:48-91
    torch.manual_seed(1001)
    E, H, V, L = 32, 64, 97, 2
    dec = ref_s2s.DecoderRNN(E, H, V, num_layers=L, rnn_type="lstm").eval()
    torch.save(dec.state_dict(), os.path.join(OUT, "ref_ckpt_seq2seq_lstm.pth"))
    f = features(4, E, 501)
    caps = captions(4, 7, V, 502)
    io["s2s_dims"] = np.array([E, H, V, L])
    io["s2s_logits"] = dec(f, caps).numpy()
    ids = [dec.sample(r, max_len=12, start_idx=2, end_idx=-1) for r in f]
    step_logits = []                      # the logits behind each sampled token, re-derived with the reference's modules
    for r, row in zip(f, ids):
        x, hidden, lg = dec.feature_fc(r.unsqueeze(0)).unsqueeze(1), None, []
        for t in row:
            out, hidden = dec.rnn(x, hidden)
            lg.append(dec.fc(out.squeeze(1))[0])
            x = dec.feature_fc(dec.embed(torch.tensor([t])).unsqueeze(1))
        step_logits.append(torch.stack(lg))
    assert torch.stack(step_logits).argmax(-1).tolist() == ids
    io["s2s_ids"], io["s2s_gaps"] = np.array(ids), gaps(torch.stack(step_logits))

    # word-level transformer - transformer_word_level.py:28-112
    torch.manual_seed(1002)
    E, FF, eh, dh, V = 64, 128, 4, 4, 97
    ref_twl.vocab_size = V
    m = ref_twl.ImageCaptioningModel(FeaturePassThrough(), ref_twl.TransformerEncoderBlock(E, FF, eh),
                                     ref_twl.TransformerDecoderBlock(E, FF, dh), embed_dim=E).eval()
    torch.save(m.state_dict(), os.path.join(OUT, "ref_ckpt_word_tfm.pth"))
    f = features(5, E, 503)
    caps = captions(5, 10, V, 504)
    io["word_dims"] = np.array([E, FF, eh, dh, V])
    io["word_logits"] = m(f, caps).numpy()
    runs = [greedy_with_reference(m, r, 10, 2, 3, False) for r in f]
    io["word_ids"], io["word_gaps"] = np.array([r[0] for r in runs]), gaps(torch.stack([r[1] for r in runs]))

    # char-level transformer - transformercharmiguel.py:30-113 (vocabulary fixed at 81 symbols by datachar)
    torch.manual_seed(1003)
    E, FF, eh, dh = 64, 64, 2, 4
    m = ref_chr.ImageCaptioningModel(FeaturePassThrough(), ref_chr.TransformerEncoderBlock(E, FF, eh),
                                     ref_chr.TransformerDecoderBlock(E, FF, dh), embed_dim=E).eval()
    torch.save(m.state_dict(), os.path.join(OUT, "ref_ckpt_char_tfm.pth"))
    f = features(3, E, 505)
    caps = captions(3, 12, 81, 506, start=0)
    io["char_dims"] = np.array([E, FF, eh, dh, 81])
    io["char_logits"] = m(f, caps).numpy()

    # Bahdanau - transformer_word_level_bahdnau.py:53-166 (BatchNorm1d in the encoder block and over the logits)
    torch.manual_seed(1004)
    E, ENC, H, A, V, DENSE = 32, 64, 64, 16, 97, 48
    m = ref_bah.ImageCaptioningModel(FeaturePassThrough(), ref_bah.TransformerEncoderBlock(ENC, DENSE, 8, dropout=0.5),
                                     ref_bah.BahdanauDecoder(E, ENC, H, A, V, dropout=0.5)).eval()
    randomise_bn(m, 77)
    torch.save(m.state_dict(), os.path.join(OUT, "ref_ckpt_bahdanau.pth"))
    f = features(4, ENC, 507)
    caps = captions(4, 9, V, 508)
    io["bah_dims"] = np.array([E, ENC, H, A, V, DENSE])
    io["bah_logits"] = m(f, caps).numpy()
    runs = [greedy_with_reference(m, r, 10, 2, 3, False) for r in f]
    io["bah_ids"], io["bah_gaps"] = np.array([r[0] for r in runs]), gaps(torch.stack([r[1] for r in runs]))

    np.savez_compressed(os.path.join(OUT, "ref_ckpt_io.npz"), **io)
    for k, v in io.items():
        print(k, v.shape)


if __name__ == "__main__":
    main()
