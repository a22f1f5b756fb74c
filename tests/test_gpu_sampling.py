"""GPU: the top-k sampling head (softmax -> top-k -> multinomial, TransformerTest.py:147-157) through the C-ABI."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
D = "cuda:0"


@pytest.mark.parametrize("rows,V,k", [(37, 6756, 10), (5, 81, 10), (3, 8192, 16), (4, 12, 3), (2, 6756, 1)])
def test_topk_probabilities_match_torch(rows, V, k):
    from image_captioning_amd import ops
    g = torch.Generator().manual_seed(rows * 1000 + V)
    logits = torch.randn(rows, V, generator=g) * 3
    want_p, want_i = torch.topk(torch.softmax(logits.double(), -1), k, dim=-1)
    ids, probs, idx = ops.topk_sample(logits.to(D), k, seed=7, offset=0)
    assert torch.equal(idx.cpu().long(), want_i)                       # random floats: no ties
    torch.testing.assert_close(probs.cpu().double(), want_p, rtol=1e-4, atol=1e-7)
    assert all(int(ids[r]) in set(want_i[r].tolist()) for r in range(rows))
    # a padded (strided) logits matrix, as the decoders hand it over
    wide = torch.full((rows, V + 4), -1e30)
    wide[:, :V] = logits
    ids2, probs2, idx2 = ops.topk_sample(wide.to(D)[:, :V], k, seed=7, offset=0)
    assert torch.equal(ids2, ids) and torch.equal(idx2, idx)


def test_ties_take_the_lower_index_and_streams_are_reproducible():
    from image_captioning_amd import ops
    logits = torch.zeros(6, 50)
    logits[:, [40, 3, 17]] = 2.0
    ids, probs, idx = ops.topk_sample(logits.to(D), 4, seed=1, offset=5)
    assert idx.cpu()[:, :3].tolist() == [[3, 17, 40]] * 6 and idx.cpu()[:, 3].tolist() == [0] * 6
    again, _, _ = ops.topk_sample(logits.to(D), 4, seed=1, offset=5)
    assert torch.equal(again, ids)
    # row r of a call at offset o draws what row 0 draws at offset o + r
    for r in range(6):
        one, _, _ = ops.topk_sample(logits[:1].to(D), 4, seed=1, offset=5 + r)
        assert int(one[0]) == int(ids[r])


def test_draw_frequencies_follow_the_renormalised_topk_probabilities():
    from image_captioning_amd import ops
    V, k, n = 97, 5, 60000
    g = torch.Generator().manual_seed(3)
    row = torch.randn(V, generator=g) * 2
    p, i = torch.topk(torch.softmax(row.double(), -1), k)
    p = (p / p.sum()).numpy()
    ids, _, _ = ops.topk_sample(row.repeat(n, 1).to(D), k, seed=11, offset=123)
    counts = np.array([(ids.cpu() == int(t)).sum().item() for t in i])
    assert counts.sum() == n                                            # nothing outside the top k
    sigma = np.sqrt(n * p * (1 - p))
    assert (np.abs(counts - n * p) < 4.5 * sigma).all(), (counts, n * p)


def test_limits_are_reported():
    from image_captioning_amd import ops
    with pytest.raises(ops.IcapError):
        ops.topk_sample(torch.zeros(2, 100, device=D), 17)
    with pytest.raises(ops.IcapError):
        ops.topk_sample(torch.zeros(2, 9000, device=D), 5)
    with pytest.raises(ops.IcapError):
        ops.topk_sample(torch.zeros(2, 100), 5)                       # CPU tensor: no fallback


@pytest.mark.parametrize("which,dims", [("word", [64, 128, 4, 4, 97]), ("char", [64, 128, 2, 4, 81])])
def test_sampling_with_k_1_is_greedy_decoding(which, dims):
    model = H.tfm_model(which, dims).to(D).eval()
    feats = torch.randn(9, dims[0], generator=torch.Generator().manual_seed(5)).to(D)
    greedy = model.greedy_ids_from_features(feats, 12)
    sampled = model.sample_ids_from_features(feats, 12, top_k=1, seed=99)
    assert torch.equal(greedy, sampled)
    a = model.sample_ids_from_features(feats, 12, top_k=10, seed=4)
    b = model.sample_ids_from_features(feats, 12, top_k=10, seed=4)
    c = model.sample_ids_from_features(feats, 12, top_k=10, seed=5)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert int(a.min()) >= 0 and int(a.max()) < dims[4]
