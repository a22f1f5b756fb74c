"""GPU: the persistent whole-caption transformer decoder (icap_greedy_tfm_persist_f32: one launch, weights resident in
LDS, {value, tag} polling between workgroups) against the reference's goldens and the step-per-launch driver."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
D = "cuda:0"


def _inputs(m, feats):
    wd = m._greedy_weights()
    enc = m.encoder.forward_independent(feats)
    ca = m.decoder.cross_attention_single(enc)
    start = torch.full((ca.shape[0],), m.start_token, device=ca.device, dtype=torch.int64)
    return wd, ca, start


@pytest.mark.parametrize("name,which", [("char_tfm_full", "char"), ("word_tfm_full", "word"), ("word_tfm_small", "word")])
def test_persistent_decoder_matches_the_reference_golden(name, which):
    from image_captioning_amd import ops
    g = H.load(name)
    E, FF, eh, dh, V, start, end, steps = [int(v) for v in g["dims"]]
    m = H.tfm_model(which, g["dims"]).to(D)
    feats = H.features(8, E, 104).to(D)
    wd, ca, st = _inputs(m, feats)
    max_nb = ops.greedy_tfm_persist_max_nb(E, FF, V)
    assert max_nb >= 1, "shape must be supported"
    ids, logits = [], []
    for s in range(0, 8, max_nb):
        i, lg = ops.greedy_tfm_persist(wd, ca[s:s + max_nb].contiguous(), st[s:s + max_nb].contiguous(), steps, want_logits=True)
        ids.append(i)
        logits.append(lg)
    ids, logits = torch.cat(ids, 0).cpu().numpy(), torch.cat(logits, 1)
    assert ids.min() >= 0, "the persistent launch reported that it could not make progress"
    H.assert_ids_match(ids, g["ids_forced"], g["gaps"], f"{name} persistent decoder")
    want = torch.from_numpy(g["logits_img0"])
    got = logits[:3, 0].cpu()
    assert ((got - want).abs() <= 1e-4 + 1e-3 * want.abs()).all(), (got - want).abs().max()
    # against the step-per-launch driver: same ids (gap rule), logits to fp32 rounding
    ids2, lg2 = ops.greedy_tfm(wd, ca, st, steps, want_logits=True)
    H.assert_ids_match(ids, ids2.cpu().numpy(), g["gaps"], f"{name} vs step driver")
    same = torch.from_numpy(ids == ids2.cpu().numpy()).all(1)
    err = (logits[:, same.to(D)] - lg2[:, same.to(D)]).abs().max().item()
    assert err <= 1e-3 * lg2.abs().max().item(), err


@pytest.mark.parametrize("NB", [1, 3, 8])
def test_persistent_decoder_repeated_and_ragged_calls(NB):
    """Consecutive launches re-use the exchange workspace pattern (fresh tags every launch), with caption counts that do
    not fill the lock-step width and step counts of 1 and 58."""
    from image_captioning_amd import ops
    g = H.load("char_tfm_full")
    E, FF, eh, dh, V, start, end, steps = [int(v) for v in g["dims"]]
    m = H.tfm_model("char", g["dims"]).to(D)
    wd, ca, st = _inputs(m, H.features(8, E, 104).to(D))
    for n_steps in (58, 1, 7):
        a, _ = ops.greedy_tfm_persist(wd, ca[:NB].contiguous(), st[:NB].contiguous(), n_steps)
        b, _ = ops.greedy_tfm_persist(wd, ca[:NB].contiguous(), st[:NB].contiguous(), n_steps)
        assert torch.equal(a, b) and a.min() >= 0
        H.assert_ids_match(a.cpu().numpy(), g["ids_forced"][:NB, :n_steps], g["gaps"][:NB, :n_steps], f"NB={NB} steps={n_steps}")


def test_persistent_decoder_argument_errors():
    from image_captioning_amd import ops
    assert ops.greedy_tfm_persist_max_nb(510, 512, 81) == 0           # E % 4
    assert ops.greedy_tfm_persist_max_nb(512, 512, 1 << 18) == 0      # vocabulary beyond the tag's index field
    assert ops.greedy_tfm_persist_max_nb(4096, 16384, 6756) == 0      # 300 MB of weights cannot live in LDS


@pytest.mark.parametrize("B", [1, 8, 37])
def test_table_walk_decoding_of_the_char_model(B):
    """The char model's default greedy path: one batched decoder pass over 81 x B rows, then look-ups - against the
    reference's goldens (B <= 8) and against the step-per-launch and GEMM paths for every image."""
    g = H.load("char_tfm_full")
    E, FF, eh, dh, V, start, end, steps = [int(v) for v in g["dims"]]
    m = H.tfm_model("char", g["dims"]).to(D)
    feats = torch.cat([H.features(8, E, 104), H.features(max(B - 8, 1), E, 105)])[:B].to(D)
    ids, logits = m.greedy_ids_from_features(feats, steps, want_logits=True)                  # mode None -> table
    ids_t, logits_t = m.greedy_ids_from_features(feats, steps, want_logits=True, mode="table")
    assert torch.equal(ids, ids_t) and ids.shape == (B, steps) and logits.shape == (steps, B, V)
    n = min(B, 8)
    H.assert_ids_match(ids[:n].cpu().numpy(), g["ids_forced"][:n], g["gaps"][:n], "table walk vs reference")
    if B >= 1:
        want = torch.from_numpy(g["logits_img0"])
        assert ((logits[:3, 0].cpu() - want).abs() <= 1e-4 + 1e-3 * want.abs()).all()
    ref_ids, ref_logits = m.greedy_ids_from_features(feats, steps, want_logits=True, mode="steps" if B <= 8 else "gemm")
    top2 = torch.topk(ref_logits, 2, dim=2).values
    gaps = (top2[..., 0] - top2[..., 1]).t().cpu().numpy()
    H.assert_ids_match(ids.cpu().numpy(), ref_ids.cpu().numpy(), gaps, "table walk vs stepwise path")
    same = (ids == ref_ids).all(1)
    assert (logits[:, same] - ref_logits[:, same]).abs().max().item() <= 1e-3 * ref_logits.abs().max().item()
    assert m.caption_ids is not None and m.greedy_ids_from_features(feats, 5).shape == (B, 5)   # V > 2 * 5: falls back to steps / gemm


def test_table_walk_kernel_alone():
    from image_captioning_amd import ops
    V, B, steps = 11, 3, 9
    g = torch.Generator().manual_seed(0)
    nxt = torch.randint(0, V, (V, B), generator=g)
    start = torch.tensor([2, 0, 10])
    tab = torch.randn(V * B, 12, generator=g)
    ids, lg = ops.table_walk(nxt.to(D), start.to(D), steps, logits_tab=tab.to(D), V=V)
    for b in range(B):
        tok = int(start[b])
        for t in range(steps):
            assert torch.equal(lg[t, b].cpu(), tab[tok * B + b, :V])
            tok = int(nxt[tok, b])
            assert int(ids[b, t]) == tok
