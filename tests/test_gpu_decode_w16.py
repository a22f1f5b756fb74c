"""GPU: the step-per-launch greedy drivers with bf16 WEIGHT STREAMING (icap_greedy_{rnn,tfm,bah}_w16_f32) - the weight
matrices are read as bf16 copies, everything else stays fp32.

Two checks per decoder:
  * the kernels are exact: on a model whose weight matrices are already bf16-representable the bf16-streaming driver and the
    fp32 driver multiply the same numbers, so their logits agree to fp32 summation-order noise and their ids are equal;
  * stated tolerance of the mode on ordinary fp32 weights, against the vectors captured from the reference's own classes
    (tests/golden): logits within 2e-2 of their largest magnitude, ids identical up to the first step whose reference
    top-2 gap is below twice the observed logit error."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

W16_LOGIT_TOL = 2e-2      # of max |logit|


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture
def ops():
    from image_captioning_amd import ops
    yield ops
    ops.set_decode_weights(None)


def round_matrices_to_bf16(model):
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 2 and "embed" not in name and "embedding" not in name:
                p.copy_(p.to(torch.bfloat16).float())


def build(kind):
    if kind in ("lstm", "gru"):
        g = H.load(f"seq2seq_{kind}_full")
        m = H.rnn_decoder(kind, g["dims"])
        feats = H.features(8, int(g["dims"][0]), 101)
        run = lambda m, f, n=20: m.sample_ids(f, n, want_logits=True)
    elif kind == "word":
        g = H.load("word_tfm_full")
        m = H.tfm_model("word", g["dims"])
        feats = H.features(8, int(g["dims"][0]), 104)
        steps = int(g["dims"][7])
        run = lambda m, f, n=steps: m.greedy_ids_from_features(f, n, want_logits=True, mode="steps")
    else:
        g = H.load("bahdanau_full")
        m = H.bah_model(g["dims"])
        feats = H.features(8, int(g["dims"][1]), 106)
        run = lambda m, f, n=20: m.greedy_ids_from_features(f, n, want_logits=True)
    return g, m, feats, run


def ids_equal_up_to_gap(got, want, gaps, floor, what):
    got, want, gaps = np.asarray(got), np.asarray(want), np.asarray(gaps)
    checked = 0
    for r in range(want.shape[0]):
        for t in range(want.shape[1]):
            if want[r, t] < 0 or gaps[r, t] < floor:
                break
            assert got[r, t] == want[r, t], f"{what}: row {r} step {t}: got {got[r, t]} want {want[r, t]} (gap {gaps[r, t]:.3e}, floor {floor:.3e})"
            checked += 1
    return checked


def top2_gaps(logits):
    """logits [steps, NB, V] -> gaps [NB, steps]"""
    top = torch.topk(logits.float(), 2, dim=-1).values
    return (top[..., 0] - top[..., 1]).t().cpu().numpy()


@pytest.mark.parametrize("kind", ["lstm", "gru", "word", "bah"])
def test_w16_driver_is_exact_on_bf16_representable_weights(ops, kind):
    _, m, feats, run = build(kind)
    round_matrices_to_bf16(m)
    m = m.to(dev())
    f = feats.to(dev())
    ops.set_decode_weights("f32")
    ids0, lg0 = run(m, f)
    ops.set_decode_weights("bf16")
    ids1, lg1 = run(m, f)
    assert lg0.shape == lg1.shape
    # same products, different summation order: only compare steps before the two decodes could have split
    gaps = top2_gaps(lg0)
    n = ids_equal_up_to_gap(ids1.cpu().numpy(), ids0.cpu().numpy(), gaps, H.GAP_EXEMPT, kind)
    assert n >= 8
    same = (ids0 == ids1).all(dim=1).cpu().numpy()
    for r in np.nonzero(same)[0]:
        a, b = lg0[:, r].double().cpu(), lg1[:, r].double().cpu()
        assert float((a - b).abs().max()) <= 1e-5 + 1e-4 * float(a.abs().max()), f"{kind} row {r}"
    assert same.sum() >= 4


@pytest.mark.parametrize("kind", ["lstm", "gru", "word", "bah"])
def test_w16_driver_within_stated_tolerance_of_the_reference(ops, kind):
    g, m, feats, run = build(kind)
    m = m.to(dev())
    f = feats.to(dev())
    ops.set_decode_weights("bf16")
    ids, lg = run(m, f)
    want = torch.from_numpy(g["logits_img0"]).double()                    # the reference's first three steps of image 0
    got = lg[:want.shape[0], 0].double().cpu()
    err = float((got - want).abs().max())
    scale = float(want.abs().max())
    assert err <= W16_LOGIT_TOL * scale, f"{kind}: logit error {err:.3e} against max |logit| {scale:.3e}"
    assert err > 0.0                                                       # the bf16 copies really were read
    # error over the whole decode, measured against this library's fp32 driver (itself pinned to the reference elsewhere)
    ops.set_decode_weights("f32")
    ids32, lg32 = run(m, f)
    same = (ids == ids32).all(dim=1)
    worst = max(float((lg[:, r] - lg32[:, r]).abs().max()) for r in range(ids.shape[0]) if bool(same[r])) if bool(same.any()) else err
    worst = max(worst, err)
    assert worst <= W16_LOGIT_TOL * float(lg32.abs().max())
    n = ids_equal_up_to_gap(ids.cpu().numpy(), g["ids_forced"], g["gaps"], max(2.0 * worst, H.GAP_EXEMPT), kind)
    assert n >= 8, f"{kind}: only {n} ids were comparable"


def test_w16_rejects_rows_that_are_not_multiples_of_8(ops):
    from image_captioning_amd import seq2seqbasic as s2s
    from image_captioning_amd.synthetic import seeded_state_dict
    m = s2s.DecoderRNN(12, 20, 40, num_layers=1, rnn_type="lstm")
    m.load_state_dict(seeded_state_dict(m, seed=5))
    m = m.eval().to(dev())
    ops.set_decode_weights("bf16")
    with pytest.raises(ops.IcapError):
        m.sample_ids(H.features(2, 12, 1).to(dev()), 4)
    ops.set_decode_weights("f32")
    assert m.sample_ids(H.features(2, 12, 1).to(dev()), 4).shape == (2, 4)
