"""GPU: the latency form of the recurrent greedy drivers (one launch per step for the Bahdanau LSTMCell decoder, two for the
two-layer LSTM / GRU; icap_greedy_bah_fused_f32 / icap_greedy_rnn2_fused_f32) against the reference's golden captions and
against the launch-per-cell drivers, at the goldens' small widths and at the widths the reference trains
(seq2seqbasic.py:194-199, transformer_word_level_bahdnau.py:183-188)."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

LOGIT_RTOL = 1e-3       # fp32 sums in another order: relative to the largest logit of the step


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _per_group(run, feats, group):
    ids, logits = [], []
    for s in range(0, feats.shape[0], group):
        i, lg = run(feats[s:s + group].contiguous())
        ids.append(i)
        logits.append(lg)
    return torch.cat(ids, 0), torch.cat(logits, 1)


def _close(a, b, what):
    a, b = a.cpu().numpy(), b.cpu().numpy()
    tol = LOGIT_RTOL * max(1.0, float(np.abs(b).max()))
    assert np.abs(a - b).max() <= tol, f"{what}: logits differ by {np.abs(a - b).max():.3e} (tolerance {tol:.3e})"


def _compare(run, feats, golden, what):
    """fused (groups of 1 and of 2) against the launch-per-cell driver on the same rows, and against the golden ids."""
    from image_captioning_amd import ops
    ops.set_decode_fused(False)
    try:
        ids_ref, lg_ref = _per_group(run, feats, 2)
    finally:
        ops.set_decode_fused(True)
    gap = None
    for group in (1, 2):
        ids, lg = _per_group(run, feats, group)
        _close(lg, lg_ref, f"{what} group {group}")
        # ids may differ from the per-cell driver's only after a step whose top-2 gap is within the logit tolerance
        top2 = lg_ref.topk(2, dim=-1).values
        gap = (top2[..., 0] - top2[..., 1]).t().cpu().numpy()                    # [rows, steps]
        H.assert_ids_match(ids.cpu().numpy(), ids_ref.cpu().numpy(), np.where(gap < 1e-3, 0.0, 1.0), f"{what} group {group} vs per-cell")
        if golden is not None:
            H.assert_ids_match(ids.cpu().numpy(), golden["ids_forced"], golden["gaps"], f"{what} group {group} vs golden")
            _close(lg[:3, 0], torch.from_numpy(golden["logits_img0"]), f"{what} group {group} golden logits")
    return ids_ref


@pytest.mark.parametrize("kind", ["lstm", "gru"])
def test_rnn2_fused_small(kind):
    g = H.load(f"seq2seq_{kind}_small")
    m = H.rnn_decoder(kind, g["dims"]).to(dev())
    feats = H.features(8, int(g["dims"][0]), 101).to(dev())
    _compare(lambda f: m.sample_ids(f, 20, want_logits=True), feats, g, f"{kind} fused")


def test_bahdanau_fused_small():
    g = H.load("bahdanau_small")
    m = H.bah_model(g["dims"]).to(dev())
    feats = H.features(8, int(g["dims"][1]), 106).to(dev())
    _compare(lambda f: m.greedy_ids_from_features(f, 20, want_logits=True), feats, g, "bahdanau fused")


@pytest.mark.parametrize("kind", ["lstm", "gru"])
def test_rnn2_fused_reference_widths(kind):
    """embed 256, hidden 512, two layers, the flickr vocabulary: ragged last vocabulary block (6756 = 422 x 16 + 4)."""
    m = H.rnn_decoder(kind, (256, 512, 6756, 2)).to(dev())
    feats = H.features(4, 256, 303).to(dev())
    ids = _compare(lambda f: m.sample_ids(f, 20, want_logits=True), feats, None, f"{kind} fused 512")
    assert len(set(ids.flatten().tolist())) > 4


def test_bahdanau_fused_reference_widths():
    m = H.bah_model((512, 512, 512, 256, 6756, 512)).to(dev())
    feats = H.features(4, 512, 304).to(dev())
    ids = _compare(lambda f: m.greedy_ids_from_features(f, 10, want_logits=True), feats, None, "bahdanau fused 512")
    assert len(set(ids.flatten().tolist())) > 2


def test_fused_paths_are_taken_and_limits_are_errors():
    """One and two captions go through the fused entry points (traced), three do not; too many captions is an error."""
    from image_captioning_amd import ops, _native as N
    g = H.load("seq2seq_lstm_small")
    m = H.rnn_decoder("lstm", g["dims"]).to(dev())
    wd = m._weights()
    x0 = torch.randn(2, int(g["dims"][1]), device=dev())
    ops.greedy_rnn(wd, x0, 5)
    assert "_fused" in wd
    with pytest.raises(N.IcapError):
        ops.greedy_rnn2_fused(wd, torch.randn(9, int(g["dims"][1]), device=dev()), 5)
    m1 = H.rnn_decoder("lstm", (32, 64, 97, 1)).to(dev())
    wd1 = m1._weights()
    ops.greedy_rnn(wd1, x0, 5)                       # one layer: the launch-per-cell driver
    assert "_fused" not in wd1


@pytest.mark.parametrize("nb", [3, 5, 8])
def test_fused_entry_points_with_wider_groups(nb):
    """The C entry points take up to 8 captions (the Python drivers route at most 2 to them): groups of 3 / 5 / 8 exercise the
    LDS-staged arg-max partials and the cell loop over more elements than threads, against the launch-per-cell drivers."""
    from image_captioning_amd import ops
    d = dev()
    for kind in ("lstm", "gru"):
        m = H.rnn_decoder(kind, (256, 512, 6756, 2)).to(d)
        wd = m._weights()
        x0 = ops.linear(H.features(nb, 256, 41 + nb).to(d), wd["ff_w"], bias=wd["ff_b"])
        ids, lg = ops.greedy_rnn2_fused(wd, x0, 12, want_logits=True)
        ops.set_decode_fused(False)
        try:
            ids_ref, lg_ref = ops.greedy_rnn(wd, x0, 12, want_logits=True)
        finally:
            ops.set_decode_fused(True)
        _close(lg, lg_ref, f"{kind} fused nb={nb}")
        top2 = lg_ref.topk(2, dim=-1).values
        gap = (top2[..., 0] - top2[..., 1]).t().cpu().numpy()
        H.assert_ids_match(ids.cpu().numpy(), ids_ref.cpu().numpy(), np.where(gap < 1e-3, 0.0, 1.0), f"{kind} fused nb={nb}")
    mb = H.bah_model((512, 512, 512, 256, 6756, 512)).to(d)
    p = mb.decoder._weights()
    ctx = mb.encoder.rows_forward(H.features(nb, 512, 51 + nb).to(d))
    start = torch.full((nb,), 2, device=d, dtype=torch.int64)
    ids, lg = ops.greedy_bah_fused(p, ctx, start, 10, want_logits=True)
    ops.set_decode_fused(False)
    try:
        ids_ref, lg_ref = ops.greedy_bah(p, ctx, start, 10, want_logits=True)
    finally:
        ops.set_decode_fused(True)
    _close(lg, lg_ref, f"bahdanau fused nb={nb}")
    top2 = lg_ref.topk(2, dim=-1).values
    gap = (top2[..., 0] - top2[..., 1]).t().cpu().numpy()
    H.assert_ids_match(ids.cpu().numpy(), ids_ref.cpu().numpy(), np.where(gap < 1e-3, 0.0, 1.0), f"bahdanau fused nb={nb}")


@pytest.mark.parametrize("kind", ["lstm", "gru"])
def test_rnn2_persistent_caption(kind):
    """The two-layer LSTM's / GRU's whole caption as one persistent launch (icap_greedy_rnn2_persist_f32): golden ids at the
    small widths, and at the reference's widths ids / logits against the launch-per-cell driver, one and two captions per
    launch (five captions: 2 + 2 + 1), twice in a row (the exchange buffers are cleared per call)."""
    from image_captioning_amd import ops
    g = H.load(f"seq2seq_{kind}_small")
    m = H.rnn_decoder(kind, g["dims"]).to(dev())
    feats = H.features(8, int(g["dims"][0]), 101).to(dev())
    _compare(lambda f: m.sample_ids(f, 20, want_logits=True, mode="persistent"), feats, g, f"{kind} persistent")
    mb = H.rnn_decoder(kind, (256, 512, 6756, 2)).to(dev())
    feats = H.features(5, 256, 303).to(dev())
    ops.set_decode_fused(False)
    try:
        ids_ref, lg_ref = mb.sample_ids(feats, 12, want_logits=True)
    finally:
        ops.set_decode_fused(True)
    top2 = lg_ref.topk(2, dim=-1).values
    gap = (top2[..., 0] - top2[..., 1]).t().cpu().numpy()
    for rep in range(2):
        ids, lg = mb.sample_ids(feats, 12, want_logits=True, mode="persistent")
        _close(lg, lg_ref, f"{kind} persistent 512 (call {rep})")
        H.assert_ids_match(ids.cpu().numpy(), ids_ref.cpu().numpy(), np.where(gap < 1e-3, 0.0, 1.0), f"{kind} persistent 512 (call {rep})")
    assert ops.greedy_rnn2_persist_max_nb(kind == "lstm", 512, 6756) == 2
    wd = mb._weights()
    with pytest.raises(ops.IcapError):
        ops.greedy_rnn2_persist(wd, torch.zeros(3, 512, device=dev()), 4)
    with pytest.raises(ops.IcapError):      # one layer has no persistent form
        H.rnn_decoder(kind, (32, 64, 97, 1)).to(dev()).sample_ids(H.features(1, 32, 7).to(dev()), 4, mode="persistent")


def test_bahdanau_persistent_caption():
    """The whole caption as one persistent launch (icap_greedy_bah_persist_f32): golden ids at the small widths, and at the
    reference's widths ids / logits against the launch-per-cell driver, one and two captions per launch, twice in a row (the
    exchange buffers are cleared per call)."""
    from image_captioning_amd import ops
    g = H.load("bahdanau_small")
    m = H.bah_model(g["dims"]).to(dev())
    feats = H.features(8, int(g["dims"][1]), 106).to(dev())
    ids, lg = m.greedy_ids_from_features(feats, 20, want_logits=True, mode="persistent")
    H.assert_ids_match(ids.cpu().numpy(), g["ids_forced"], g["gaps"], "bahdanau persistent vs golden")
    _close(lg[:3, 0], torch.from_numpy(g["logits_img0"]), "bahdanau persistent golden logits")
    mb = H.bah_model((512, 512, 512, 256, 6756, 512)).to(dev())
    feats = H.features(5, 512, 304).to(dev())
    ops.set_decode_fused(False)
    try:
        ids_ref, lg_ref = mb.greedy_ids_from_features(feats, 12, want_logits=True)
    finally:
        ops.set_decode_fused(True)
    top2 = lg_ref.topk(2, dim=-1).values
    gap = (top2[..., 0] - top2[..., 1]).t().cpu().numpy()
    for rep in range(2):
        ids, lg = mb.greedy_ids_from_features(feats, 12, want_logits=True, mode="persistent")
        _close(lg, lg_ref, f"bahdanau persistent 512 (call {rep})")
        H.assert_ids_match(ids.cpu().numpy(), ids_ref.cpu().numpy(), np.where(gap < 1e-3, 0.0, 1.0), f"bahdanau persistent 512 (call {rep})")
    p = mb.decoder._weights()
    assert ops.greedy_bah_persist_max_nb(512, 6756) == 2
    with pytest.raises(ops.IcapError):
        ops.greedy_bah_persist(p, torch.zeros(3, 512, device=dev()), torch.full((3,), 2, device=dev(), dtype=torch.int64), 4)
