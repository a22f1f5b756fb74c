"""GPU parity of the word / char transformer captioners' training step against the reference goldens (dropout 0) and
against the CPU oracle with shared dropout masks; attention-gradient kernel against torch autograd."""
import math

import pytest
import torch

from oracle import torch_ref as O
from oracle import train_ref as TR
from tests import helpers as H

pytestmark = pytest.mark.gpu
D = "cuda:0"


@pytest.mark.parametrize("R,Bk,E,heads", [(60, 12, 512, 8), (7, 7, 64, 4), (174, 3, 512, 2), (5, 1, 128, 4)])
def test_attention_core_gradient(R, Bk, E, heads):
    from image_captioning_amd import train_ops as T
    g = torch.Generator().manual_seed(R + Bk)
    q = torch.randn(R, E, generator=g, requires_grad=True)
    k = torch.randn(Bk, E, generator=g, requires_grad=True)
    v = torch.randn(Bk, E, generator=g, requires_grad=True)
    d = E // heads
    qh, kh, vh = (t.view(-1, heads, d).transpose(0, 1) for t in (q, k, v))
    out = (torch.softmax(qh @ kh.transpose(1, 2) / math.sqrt(d), -1) @ vh).transpose(0, 1).reshape(R, E)
    dO = torch.randn(R, E, generator=g)
    out.backward(dO)
    dq, dk, dv = T.mha_bwd(q.detach().to(D), k.detach().to(D), v.detach().to(D), dO.to(D), heads)
    torch.testing.assert_close(dq.cpu(), q.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(dk.cpu(), k.grad, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(dv.cpu(), v.grad, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("name", ["word", "char"])
def test_training_step_matches_reference_golden(name):
    from image_captioning_amd.training import compute_loss
    g = H.load(f"{name}_tfm_train_small")
    E, FF, eh, dh, V, B, T, pad = [int(v) for v in g["dims"]]
    model = H.tfm_model(name, [E, FF, eh, dh, V], seed=41, dropout=0.0).to(D).train()
    feats = torch.randn(B, E, generator=torch.Generator().manual_seed(701)).to(D).requires_grad_(True)
    caps = torch.from_numpy(g["caps"]).to(D)
    pred = model.forward_features(feats, caps)
    loss = compute_loss(pred[:, :-1], caps[:, 1:], pad)
    loss.backward()
    torch.testing.assert_close(pred.detach().cpu(), torch.from_numpy(g["logits"]), rtol=1e-3, atol=1e-4)
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    torch.testing.assert_close(feats.grad.cpu(), torch.from_numpy(g["grad_feats"]), rtol=5e-3, atol=1e-6)
    for n, p in model.named_parameters():
        want = torch.from_numpy(g["grad." + n])
        if want.numel() == 0:
            assert p.grad is None, n
            continue
        torch.testing.assert_close(p.grad.cpu(), want, rtol=5e-3, atol=2e-6, msg=lambda m: f"{n}: {m}")


@pytest.mark.parametrize("name,dims,B,T", [("word", (512, 2048, 8, 8, 1000), 12, 6), ("char", (512, 512, 2, 4, 81), 4, 58)])
def test_training_step_with_dropout_matches_oracle(name, dims, B, T):
    from image_captioning_amd.training import compute_loss
    E, FF, eh, dh, V = dims
    p = 0.3
    pad = 0 if name == "word" else 2
    model = H.tfm_model(name, dims, seed=43, dropout=p)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(D).train()
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(B, E, generator=g)
    caps = torch.randint(3, V, (B, T), generator=g)
    caps[0, T - 2:] = pad

    def mk(*shape):
        return (torch.rand(*shape, generator=g) >= p).float() / (1 - p)
    masks = {"enc_att": mk(B, E), "enc_fc1": mk(B, FF), "enc_fc2": mk(B, E), "dec_att": mk(T, B, E), "dec_fc1": mk(T, B, FF),
             "dec_fc2": mk(T, B, E)}
    sdr = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    fr = feats.clone().requires_grad_(True)
    pred_r = TR.tfm_train_forward(sdr, fr, caps, eh, dh, masks)
    loss_r = O.compute_loss(pred_r[:, :-1], caps[:, 1:], pad)
    loss_r.backward()
    model.dropout_masks = {k: v.to(D) for k, v in masks.items()}
    fg = feats.to(D).requires_grad_(True)
    pred = model.forward_features(fg, caps.to(D))
    loss = compute_loss(pred[:, :-1], caps[:, 1:].to(D), pad)
    loss.backward()
    torch.testing.assert_close(pred.detach().cpu(), pred_r.detach(), rtol=1e-3, atol=5e-4)
    assert abs(loss.item() - loss_r.item()) < 1e-4
    torch.testing.assert_close(fg.grad.cpu(), fr.grad, rtol=1e-2, atol=1e-6)
    for n, prm in model.named_parameters():
        want = sdr[n].grad
        if want is None:
            assert prm.grad is None, n
            continue
        err = (prm.grad.cpu() - want).abs().max().item()
        assert err <= 1e-2 * want.abs().max().item() + 1e-7, (n, err, want.abs().max().item())
