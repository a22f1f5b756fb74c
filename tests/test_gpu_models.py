"""GPU: the module classes (HIP path) against the vectors captured from the reference's own classes
(tests/golden, made by oracle/make_golden.py) and, for the CNN encoders, against the CPU oracle."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-3, 1e-4   # north-star tolerance for fp32 features / logits


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def close(got, want, what, rtol=RTOL, atol=ATOL):
    got = torch.as_tensor(np.asarray(got.detach().cpu() if torch.is_tensor(got) else got)).double()
    want = torch.as_tensor(np.asarray(want)).double()
    assert got.shape == want.shape, f"{what}: {tuple(got.shape)} vs {tuple(want.shape)}"
    err = (got - want).abs() - (atol + rtol * want.abs())
    assert err.max() <= 0, f"{what}: max abs diff {(got - want).abs().max():.3e}, max |want| {want.abs().max():.3e}"


# ------------------------------------------------------------------ seq2seq decoder
@pytest.mark.parametrize("kind", ["lstm", "gru"])
@pytest.mark.parametrize("tag", ["small", "full"])
def test_seq2seq_decoder(kind, tag):
    g = H.load(f"seq2seq_{kind}_{tag}")
    m = H.rnn_decoder(kind, g["dims"]).to(dev())
    E = int(g["dims"][0])
    feats = H.features(8, E, 101).to(dev())
    ids, logits = m.sample_ids(feats, 20, want_logits=True)              # all 8 in lock-step
    H.assert_ids_match(ids.cpu().numpy(), g["ids_forced"], g["gaps"], f"{kind} {tag} lock-step")
    close(logits[:3, 0], g["logits_img0"], "step logits")
    for r in range(3):                                                    # the reference's per-image API
        want = [int(v) for v in g["ids_end"][r] if v >= 0]
        if g["gaps"][r][:len(want)].min() >= H.GAP_EXEMPT:
            assert m.sample(feats[r], max_len=20, start_idx=2, end_idx=3) == want
    caps = torch.from_numpy(g["fwd_captions"]).to(dev())
    close(m(feats[:caps.shape[0]], caps), g["fwd_logits"], "teacher-forced logits")


def test_seq2seq_encoder_head():
    from image_captioning_amd import seq2seqbasic as s2s
    from image_captioning_amd.synthetic import seeded_state_dict
    import torch.nn as nn
    head = nn.Sequential()
    head.fc = nn.Linear(512, 256)
    head.bn = nn.BatchNorm1d(256, momentum=0.01)
    sd = seeded_state_dict(head, seed=12)
    enc = s2s.EncoderCNN(256, "resnet18")
    enc.load_state_dict({**enc.state_dict(), **sd})
    enc = enc.eval().to(dev())
    enc.cnn = H.FeaturePassThrough()
    close(enc(H.features(5, 512, 103).to(dev())), H.load("seq2seq_encoder_head")["out"], "fc + BatchNorm1d")


# ------------------------------------------------------------------ word / char transformer
@pytest.mark.parametrize("name,which", [("word_tfm_small", "word"), ("word_tfm_full", "word"), ("char_tfm_full", "char")])
def test_transformer(name, which):
    g = H.load(name)
    E, FF, eh, dh, V, start, end, steps = [int(v) for v in g["dims"]]
    m = H.tfm_model(which, g["dims"]).to(dev())
    feats = H.features(8, E, 104).to(dev())
    ids, logits = m.greedy_ids_from_features(feats, steps, want_logits=True)             # weight-streaming path
    H.assert_ids_match(ids.cpu().numpy(), g["ids_forced"], g["gaps"], name + " lock-step")
    close(logits[:3, 0], g["logits_img0"], "step logits")
    ids2, logits2 = m.greedy_ids_from_features(feats, steps, want_logits=True, lockstep=0)   # GEMM path
    H.assert_ids_match(ids2.cpu().numpy(), g["ids_forced"], g["gaps"], name + " gemm")
    close(logits2[:3, 0], g["logits_img0"], "step logits (gemm path)")
    for key in [k for k in g if k.startswith("fwd_caps_")]:
        shape = key[len("fwd_caps_"):]
        caps = torch.from_numpy(g[key]).to(dev())
        B = caps.shape[0]
        ff = feats[:B] if B <= 8 else H.features(B, E, 105).to(dev())
        close(m(ff, caps), g["fwd_logits_" + shape], "forward " + shape)
    close(m.encoder(feats[:3].unsqueeze(0)), g["encoder_out_3"], "encoder block")
    want = [int(v) for v in g["ids_end"][0] if v >= 0]
    if g["gaps"][0][:len(want) + 1].min() >= H.GAP_EXEMPT:
        from image_captioning_amd._tfm_common import cut_before_end
        assert cut_before_end(ids[0].tolist(), end) == want


def test_word_transformer_at_the_test_script_widths():
    """The configuration the reference's own test script builds (TransformerTest.py:328-340: EMBED_DIM 512, FF_DIM 512,
    8 / 8 heads, full vocabulary; 10 tokens, TransformerTest.py:126) against vectors from the reference's classes."""
    g = H.load("word_tfm_testscript")
    E, FF, eh, dh, V, start, end, steps = [int(v) for v in g["dims"]]
    m = H.tfm_model("word", g["dims"], seed=int(g["seed"])).to(dev())
    feats = H.features(8, E, 104).to(dev())
    for lock, what in ((None, "lock-step"), (0, "gemm")):
        ids, logits = m.greedy_ids_from_features(feats, steps, want_logits=True, lockstep=lock)
        H.assert_ids_match(ids.cpu().numpy(), g["ids_forced"], g["gaps"], "test-script widths, " + what)
        close(logits[:3, 0], g["logits_img0"], "step logits, " + what)
    caps = torch.from_numpy(g["fwd_caps_3x4"]).to(dev())
    close(m(feats[:3], caps), g["fwd_logits_3x4"], "forward 3x4")
    caps = torch.from_numpy(g["fwd_caps_12x10"]).to(dev())
    logits = m(H.features(12, E, 105).to(dev()), caps).cpu()
    close(logits[:, :, torch.from_numpy(g["fwd_logits_12x10_cols"])], g["fwd_logits_12x10"], "forward 12x10 (sampled columns)")
    sure = g["fwd_gap_12x10"] >= H.GAP_EXEMPT
    assert (logits.argmax(-1).numpy() == g["fwd_argmax_12x10"])[sure].all()
    close(m.encoder(feats[:3].unsqueeze(0)), g["encoder_out_3"], "encoder block")
    want = [int(v) for v in g["ids_end"][0] if v >= 0]
    if g["gaps"][0][:len(want) + 1].min() >= H.GAP_EXEMPT:
        from image_captioning_amd._tfm_common import cut_before_end
        assert cut_before_end(ids[0].tolist(), end) == want


# ------------------------------------------------------------------ Bahdanau
@pytest.mark.parametrize("tag", ["small", "full"])
def test_bahdanau(tag):
    g = H.load(f"bahdanau_{tag}")
    E, ENC, Hd, A, V, DENSE = [int(v) for v in g["dims"]]
    m = H.bah_model(g["dims"]).to(dev())
    feats = H.features(8, ENC, 106).to(dev())
    ids, logits = m.greedy_ids_from_features(feats, 20, want_logits=True)
    H.assert_ids_match(ids.cpu().numpy(), g["ids_forced"], g["gaps"], "bahdanau lock-step")
    close(logits[:3, 0], g["logits_img0"], "step logits")
    ids2, logits2 = m.greedy_ids_from_features(feats, 20, want_logits=True, lockstep=0)
    H.assert_ids_match(ids2.cpu().numpy(), g["ids_forced"], g["gaps"], "bahdanau gemm")
    close(logits2[:3, 0], g["logits_img0"], "step logits (gemm path)")
    for key in [k for k in g if k.startswith("fwd_caps_")]:
        shape = key[len("fwd_caps_"):]
        caps = torch.from_numpy(g[key]).to(dev())
        B = caps.shape[0]
        ff = feats[:B] if B <= 8 else H.features(B, ENC, 107).to(dev())
        close(m(ff, caps), g["fwd_logits_" + shape], "forward " + shape)
    close(m.encoder(feats[:3].unsqueeze(1)), g["encoder_out_3"], "encoder block")
    alpha, ctx = m.decoder.attention(m.encoder(feats[:5].unsqueeze(1)), H.features(5, Hd, 108).to(dev()))
    close(alpha, g["att_alpha_5"], "alpha")
    close(ctx, g["att_ctx_5"], "context")


# ------------------------------------------------------------------ CNN encoders vs the CPU oracle (parity unpinned by the reference)
def food_images():
    from image_captioning_amd.preprocess import to_model_input
    return to_model_input(H.load("food_test_u8")["images"])


@pytest.mark.parametrize("cnn_type", ["resnet18", "vgg16"])
def test_seq2seq_encoder(cnn_type):
    from image_captioning_amd import seq2seqbasic as s2s
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_images
    from oracle import backbones_ref as BR
    enc = s2s.EncoderCNN(256, cnn_type)
    enc.load_state_dict(seeded_state_dict(enc, seed=41))
    enc.eval()
    sd = {"encoder." + k: v for k, v in H.cpu_sd(enc).items()}
    imgs = torch.cat([food_images()[:3], synthetic_images(1)])
    want = BR.seq2seq_encoder(sd, imgs, cnn_type)
    close(enc.to(dev())(imgs.to(dev())), want, f"EncoderCNN {cnn_type}", atol=1e-3 * float(want.abs().max()))


def test_resnet18_encoder_at_256_pixels():
    """seq2seqtest.py:178 resizes to 256 x 256: the adaptive average pool makes any size legal (stage sizes 64 / 32 / 16 / 8)."""
    from image_captioning_amd import seq2seqbasic as s2s
    from image_captioning_amd.preprocess import resize_normalize
    from image_captioning_amd.synthetic import seeded_state_dict
    from oracle import backbones_ref as BR
    enc = s2s.EncoderCNN(256, "resnet18")
    enc.load_state_dict(seeded_state_dict(enc, seed=41))
    enc.eval()
    sd = {"encoder." + k: v for k, v in H.cpu_sd(enc).items()}
    imgs = resize_normalize(H.load("food_test_u8")["images"][:3], size=256).cpu()      # the food photos again, at 256 x 256
    assert imgs.shape == (3, 3, 256, 256)
    want = BR.seq2seq_encoder(sd, imgs, "resnet18")
    close(enc.to(dev())(imgs.to(dev())), want, "EncoderCNN resnet18 @256", atol=1e-3 * float(want.abs().max()))


def test_efficientnet_cnn_model():
    from image_captioning_amd.backbones import get_cnn_model
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_images
    from oracle import backbones_ref as BR
    cnn = get_cnn_model()
    cnn.load_state_dict(seeded_state_dict(cnn, seed=42))
    cnn.eval()
    sd = {"cnn_model." + k: v for k, v in H.cpu_sd(cnn).items()}
    imgs = torch.cat([food_images(), synthetic_images(2)])
    want = BR.cnn_model(sd, imgs)
    close(cnn.to(dev())(imgs.to(dev())), want, "EfficientNet-B0 + Linear", atol=1e-3 * float(want.abs().max()))


# ------------------------------------------------------------------ end to end: food_test/1-8.jpg -> greedy token ids
def oracle_gaps(logits):
    t = torch.topk(logits, 2, dim=-1).values
    return (t[..., 0] - t[..., 1]).numpy()


@pytest.mark.parametrize("cnn_type,rnn_type", [("resnet18", "lstm"), ("resnet18", "gru"), ("vgg16", "gru")])
def test_end_to_end_seq2seq_food(cnn_type, rnn_type):
    """The three (encoder, decoder) combinations the reference trains (seq2seqbasic.py:248-250), image -> greedy ids on
    food_test/1-8.jpg: HIP encoder + DecoderRNN.sample_ids against the CPU oracle's encoder + sample loop."""
    from image_captioning_amd import seq2seqbasic as s2s
    from image_captioning_amd.synthetic import seeded_state_dict
    from oracle import backbones_ref as BR, torch_ref as R
    V = 6756
    m = s2s.ImageCaptioningModel(256, 512, V, cnn_type=cnn_type, rnn_type=rnn_type, num_layers=2)
    m.load_state_dict(seeded_state_dict(m, seed=51))
    m.eval()
    sd = H.cpu_sd(m)
    imgs = food_images()
    feats = BR.seq2seq_encoder(sd, imgs, cnn_type)
    dsd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    assert R.rnn_kind(dsd) == rnn_type
    want, gaps = [], []
    for f in feats:
        ids, lg = R.seq2seq_sample(dsd, f, 20, 3, False)
        want.append(ids)
        gaps.append(oracle_gaps(lg))
    m = m.to(dev())
    feats_gpu = m.encoder(imgs.to(dev()))
    close(feats_gpu, feats, f"EncoderCNN {cnn_type} features", atol=1e-3 * float(feats.abs().max()))
    got = m.decoder.sample_ids(feats_gpu, 20)
    H.assert_ids_match(got.cpu().numpy(), np.array(want), np.array(gaps), f"{cnn_type}+{rnn_type} food_test")
    # the reference's own per-image entry point (DecoderRNN.sample: list of ids, stops at <END>) on the first photo
    ids0, lg0 = R.seq2seq_sample(dsd, feats[0], 20, 3, True)
    if len(ids0) == 0 or oracle_gaps(lg0)[:len(ids0)].min() >= H.GAP_EXEMPT:
        assert m.decoder.sample(feats_gpu[0], max_len=20, start_idx=2, end_idx=3) == [int(v) for v in ids0]


@pytest.mark.parametrize("which", ["word", "char", "bahdanau"])
def test_end_to_end_efficientnet_food(which):
    from image_captioning_amd import transformer_word_level as twl, transformer_word_level_bahdnau as bah, transformercharmiguel as chr_
    from image_captioning_amd.synthetic import seeded_state_dict
    from oracle import backbones_ref as BR, torch_ref as R
    V = 6756
    if which == "word":
        m, steps = twl.build_model(V), 20
    elif which == "char":
        m, steps = chr_.build_model(), 58
    else:
        m, steps = bah.build_model(V), 20
    m.load_state_dict(seeded_state_dict(m, seed=52))
    m.eval()
    sd = H.cpu_sd(m)
    imgs = food_images()
    feats = BR.cnn_model(sd, imgs)
    want, gaps = [], []
    for f in feats:
        if which == "word":
            ids, lg = R.tfm_greedy(sd, f, 8, 8, steps, 2, 3, False)
        elif which == "char":
            ids, lg = R.tfm_greedy(sd, f, 2, 4, steps, 0, 1, False)
        else:
            ids, lg = R.bah_greedy(sd, f, steps, 2, 3, False)
        want.append(ids)
        gaps.append(oracle_gaps(lg))
    m = m.to(dev())
    got = m.greedy_ids_from_features(m.cnn_model(imgs.to(dev())), steps)
    H.assert_ids_match(got.cpu().numpy(), np.array(want), np.array(gaps), f"efficientnet+{which} food_test")
