"""GPU: checkpoints written by the REFERENCE's classes (tests/golden/ref_ckpt_*.pth, made by oracle/make_golden_ckpt.py
with torch.save(model.state_dict(), path) as the reference does) load through checkpoint.load_checkpoint(strict=True) into
this build's modules and reproduce the reference's stored eval-mode logits and greedy ids."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
D = "cuda:0"
RTOL = 1e-3


def _close(got, want, what):
    got, want = got.detach().cpu().double(), torch.from_numpy(want).double()
    assert got.shape == want.shape, what
    assert ((got - want).abs() - (1e-4 + RTOL * want.abs())).max() <= 0, f"{what}: max diff {(got - want).abs().max():.3e}"


def _load(model, name):
    from image_captioning_amd.checkpoint import load_checkpoint
    res = load_checkpoint(model, os.path.join(H.GOLDEN, name), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return model.eval().to(D)


def test_seq2seq_lstm_checkpoint():
    from image_captioning_amd import seq2seqbasic as s2s
    io = H.load("ref_ckpt_io")
    E, Hh, V, L = [int(v) for v in io["s2s_dims"]]
    m = _load(s2s.DecoderRNN(E, Hh, V, num_layers=L, rnn_type="lstm"), "ref_ckpt_seq2seq_lstm.pth")
    f = H.features(4, E, 501).to(D)
    caps = torch.from_numpy(np.asarray(_captions(4, 7, V, 502))).to(D)
    _close(m(f, caps), io["s2s_logits"], "seq2seq teacher-forced logits")
    H.assert_ids_match(m.sample_ids(f, 12).cpu().numpy(), io["s2s_ids"], io["s2s_gaps"], "seq2seq greedy ids")


def _captions(B, T, V, seed, start=2):
    g = torch.Generator().manual_seed(seed)
    c = torch.randint(4, V, (B, T), generator=g)
    c[:, 0] = start
    return c.numpy()


@pytest.mark.parametrize("which", ["word", "char"])
def test_transformer_checkpoint(which):
    io = H.load("ref_ckpt_io")
    dims = io[f"{which}_dims"]
    E, V = int(dims[0]), int(dims[4])
    m = H.tfm_model(which, dims)                       # seeded weights, replaced by the checkpoint's below
    m = _load(m, f"ref_ckpt_{which}_tfm.pth")
    n, T, fs, cs, start = (5, 10, 503, 504, 2) if which == "word" else (3, 12, 505, 506, 0)
    f = H.features(n, E, fs).to(D)
    caps = torch.from_numpy(_captions(n, T, V, cs, start)).to(D)
    _close(m(f, caps), io[f"{which}_logits"], f"{which} transformer logits")
    if which == "word":
        H.assert_ids_match(m.greedy_ids_from_features(f, 10).cpu().numpy(), io["word_ids"], io["word_gaps"], "word greedy ids")


def test_bahdanau_checkpoint():
    io = H.load("ref_ckpt_io")
    m = _load(H.bah_model(io["bah_dims"]), "ref_ckpt_bahdanau.pth")
    ENC, V = int(io["bah_dims"][1]), int(io["bah_dims"][4])
    f = H.features(4, ENC, 507).to(D)
    caps = torch.from_numpy(_captions(4, 9, V, 508)).to(D)
    _close(m(f, caps), io["bah_logits"], "Bahdanau logits")
    H.assert_ids_match(m.greedy_ids_from_features(f, 10).cpu().numpy(), io["bah_ids"], io["bah_gaps"], "Bahdanau greedy ids")
    assert int(m.decoder.fc[1].num_batches_tracked) == 17            # buffers travel too


def test_save_checkpoint_round_trips_into_a_reference_shaped_state_dict(tmp_path):
    """save_checkpoint writes what the reference's torch.save(model.state_dict()) writes: same keys, same shapes, CPU tensors."""
    from image_captioning_amd.checkpoint import save_checkpoint
    io = H.load("ref_ckpt_io")
    m = _load(H.bah_model(io["bah_dims"]), "ref_ckpt_bahdanau.pth")
    save_checkpoint(m, str(tmp_path / "x.pth"))
    a = torch.load(os.path.join(H.GOLDEN, "ref_ckpt_bahdanau.pth"), weights_only=True)
    b = torch.load(str(tmp_path / "x.pth"), weights_only=True)
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert a[k].shape == b[k].shape and b[k].device.type == "cpu" and torch.equal(a[k], b[k]), k
