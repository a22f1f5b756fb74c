"""TEST INFRASTRUCTURE - not imported by the product.  Golden vectors of the word-level transformer at the widths of the
reference's own test script (TransformerTest.py:328-340: EMBED_DIM 512, FF_DIM 512, 8 / 8 heads, the full 6 756-word
vocabulary; generate_caption's max_len 10, TransformerTest.py:126), made by the reference's classes imported from
/root/reference exactly as oracle/make_golden.py does (stubbed third-party imports, seeded state_dict loaded with
strict=True).  Kept apart from make_golden.py so that making this fixture rewrites no other one.

    python oracle/make_golden_width.py        ->  tests/golden/word_tfm_testscript.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.make_golden import (OUT, REF, FeaturePassThrough, captions, features, gaps, greedy_with_reference,  # noqa: E402
                                install_stubs, pad_ids)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    import transformer_word_level as ref_twl

    from image_captioning_amd import transformer_word_level as my_twl
    from image_captioning_amd.synthetic import seeded_state_dict

    torch.set_grad_enabled(False)
    E, FF, eh, dh, V, start, end, steps = 512, 512, 8, 8, 6756, 2, 3, 10
    ref_twl.vocab_size = V
    my_twl.vocab_size = V
    mine = my_twl.ImageCaptioningModel(FeaturePassThrough(), my_twl.TransformerEncoderBlock(E, FF, eh),
                                       my_twl.TransformerDecoderBlock(E, FF, dh), embed_dim=E)
    sd = seeded_state_dict(mine, seed=23)
    ref = ref_twl.ImageCaptioningModel(FeaturePassThrough(), ref_twl.TransformerEncoderBlock(E, FF, eh),
                                       ref_twl.TransformerDecoderBlock(E, FF, dh), embed_dim=E)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    feats = features(8, E, 104)
    res = {"dims": np.array([E, FF, eh, dh, V, start, end, steps]), "seed": np.array(23)}
    ids_end, ids_forced, lg_all = [], [], []
    for f in feats:
        i1, _ = greedy_with_reference(ref, f, steps, start, end, True)
        i2, lg = greedy_with_reference(ref, f, steps, start, end, False)
        ids_end.append(i1)
        ids_forced.append(i2)
        lg_all.append(lg)
    lg_all = torch.stack(lg_all)
    res.update(ids_end=pad_ids(ids_end, steps), ids_forced=np.array(ids_forced), gaps=gaps(lg_all), logits_img0=lg_all[0, :3].numpy())
    for (B, T) in [(3, 4), (12, 10)]:            # 12 x 10: the reference's loader batch and caption length
        caps = captions(B, T, V, 300 + B, start)
        logits = ref(feats[:B] if B <= 8 else features(B, E, 105), caps)
        res[f"fwd_caps_{B}x{T}"] = caps.numpy()
        if B * T * V * 4 > (1 << 20):            # keep the fixture small: every row, a strided sample of the vocabulary
            res[f"fwd_logits_{B}x{T}_cols"] = np.arange(0, V, 17)
            res[f"fwd_logits_{B}x{T}"] = logits[:, :, ::17].numpy()
            res[f"fwd_argmax_{B}x{T}"] = logits.argmax(-1).numpy()
            top2 = torch.topk(logits, 2, dim=-1).values
            res[f"fwd_gap_{B}x{T}"] = (top2[..., 0] - top2[..., 1]).numpy()
        else:
            res[f"fwd_logits_{B}x{T}"] = logits.numpy()
    res["encoder_out_3"] = ref.encoder(feats[:3].unsqueeze(0)).numpy()
    np.savez_compressed(os.path.join(OUT, "word_tfm_testscript.npz"), **res)
    print("word_tfm_testscript ids", ids_end[0][:6], "min gap", float(gaps(lg_all).min()))


if __name__ == "__main__":
    main()
