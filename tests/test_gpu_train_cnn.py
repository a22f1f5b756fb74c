"""GPU parity of the EfficientNet-B0 (+ Linear) training-mode forward and hand-written backward against the CPU
oracle's train-mode restatement with torch autograd (parity unpinned for the backbone: the efficientnet_pytorch
package is absent, see oracle/backbones_ref.py), and of the CNN-specific gradient kernels against torch."""
import pytest
import torch
import torch.nn.functional as F

from oracle import backbones_ref as BR

pytestmark = pytest.mark.gpu
D = "cuda:0"


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed + sum(shape)))


@pytest.mark.parametrize("B,H,W,C,K,S", [(2, 14, 14, 48, 3, 1), (2, 15, 15, 24, 5, 2), (3, 8, 8, 16, 3, 2), (1, 7, 7, 64, 5, 1),
                                         (2, 14, 14, 24, 5, 2), (2, 28, 28, 40, 3, 2), (2, 10, 14, 8, 5, 2), (3, 6, 22, 12, 3, 2),
                                         (1, 56, 56, 144, 5, 2)])
def test_depthwise_gradients(B, H, W, C, K, S):
    from image_captioning_amd import train_ops as T
    from image_captioning_amd.backbones import same_pad
    lo, hi = same_pad(K, S)
    x = rnd(B, C, H, W, seed=1).requires_grad_(True)
    w = rnd(C, 1, K, K, seed=2).requires_grad_(True)
    y = F.conv2d(F.pad(x, (lo, hi, lo, hi)), w, stride=S, groups=C)
    Ho, Wo = y.shape[2], y.shape[3]
    dy = rnd(B, C, Ho, Wo, seed=3)
    y.backward(dy)
    dy2 = dy.permute(0, 2, 3, 1).contiguous().view(-1, C).to(D)
    x2 = x.detach().permute(0, 2, 3, 1).contiguous().view(-1, C).to(D)
    w2 = w.detach().view(C, K * K).t().contiguous().to(D)
    dx = T.dwconv_bwd_data(dy2, w2, B, H, W, C, K, S, lo, lo, Ho, Wo).view(B, H, W, C).permute(0, 3, 1, 2).cpu()
    dw = T.dwconv_bwd_weight(dy2, x2, B, H, W, C, K, S, lo, lo, Ho, Wo).t().reshape(C, 1, K, K).cpu()
    torch.testing.assert_close(dx, x.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dw, w.grad, rtol=1e-4, atol=1e-4)


def test_channel_affine_and_image_dot():
    from image_captioning_amd import train_ops as T
    B, HW, C = 3, 49, 32
    x, g, a = rnd(B * HW, C, seed=4), rnd(B, C, seed=5), rnd(B, C, seed=6)
    y = T.chan_affine(x.to(D), g.to(D), a.to(D), 0.5, B, HW, C).cpu()
    want = x.view(B, HW, C) * g[:, None, :] + 0.5 * a[:, None, :]
    torch.testing.assert_close(y.view(B, HW, C), want)
    d = T.image_dot(x.to(D), y.to(D), B, HW, C).cpu()
    torch.testing.assert_close(d, (x.view(B, HW, C) * y.view(B, HW, C)).sum(1), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,HW,C", [(3, 49, 32), (5, 196, 144), (2, 3136, 24), (4, 50, 1152), (7, 33, 40)])
def test_act_pool_matches_separate_passes(B, HW, C):
    """BatchNorm apply + swish + per-image mean in one pass against torch on the same numbers."""
    from image_captioning_amd import train_ops as T
    from image_captioning_amd.ops import ACT_NONE, ACT_SWISH
    x, s, b = rnd(B * HW, C, seed=7), rnd(C, seed=8), rnd(C, seed=9)
    for act, fn in ((ACT_SWISH, F.silu), (ACT_NONE, lambda v: v)):
        y, pooled = T.act_pool(x.to(D), s.to(D), b.to(D), act, B, HW)
        want = fn(x.double() * s.double() + b.double())
        torch.testing.assert_close(y.cpu().double(), want, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(pooled.cpu().double(), want.view(B, HW, C).mean(1), rtol=1e-5, atol=1e-6)
        y2, pooled2 = T.act_pool(x.to(D), s.to(D), b.to(D), act, B, HW)
        assert torch.equal(pooled, pooled2) and torch.equal(y, y2)             # fixed summation order


@pytest.mark.parametrize("B,HW,C", [(3, 49, 32), (5, 196, 144), (2, 3136, 24), (4, 50, 1152), (7, 33, 40)])
def test_gated_activation_and_its_gate_gradient_without_the_stored_activation(B, HW, C):
    """act_pool(write=False) + act_gate write swish(bn(z)) * gate straight from z, and image_dot_act takes the gate's gradient
    from z: the same numbers as the forms that go through a stored activation."""
    from image_captioning_amd import train_ops as T
    from image_captioning_amd.ops import ACT_SWISH
    z, s, b = rnd(B * HW, C, seed=11), rnd(C, seed=12), rnd(C, seed=13)
    gate, dy = torch.sigmoid(rnd(B, C, seed=14)), rnd(B * HW, C, seed=15)
    zg, sg, bg, gg = z.to(D), s.to(D), b.to(D), gate.to(D)
    a1, pooled = T.act_pool(zg, sg, bg, ACT_SWISH, B, HW)
    none, pooled2 = T.act_pool(zg, sg, bg, ACT_SWISH, B, HW, write=False)
    assert none is None and torch.equal(pooled, pooled2)
    a2 = T.act_gate(zg, sg, bg, ACT_SWISH, gg, B, HW)
    want = F.silu(z.double() * s.double() + b.double()).view(B, HW, C) * gate.double()[:, None, :]
    torch.testing.assert_close(a2.cpu().double().view(B, HW, C), want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a2, T.chan_affine(a1, gg, None, 0.0, B, HW, C), rtol=1e-6, atol=1e-7)
    dgate = T.image_dot_act(dy.to(D), zg, sg, bg, ACT_SWISH, B, HW, C)
    torch.testing.assert_close(dgate, T.image_dot(dy.to(D), a1, B, HW, C), rtol=1e-5, atol=1e-5)
    ref = (dy.double() * F.silu(z.double() * s.double() + b.double())).view(B, HW, C).sum(1)
    torch.testing.assert_close(dgate.cpu().double(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,HW,C", [(3, 49, 32), (5, 196, 144), (2, 3136, 24), (4, 50, 1152), (9, 33, 40)])
def test_gated_batchnorm_gradient_matches_the_two_step_form(B, HW, C):
    """The squeeze-excite gate applied inside the BatchNorm-gradient passes against chan_affine followed by the plain
    BatchNorm gradient (same kernels, the gated tensor written out), and against torch autograd in float64."""
    from image_captioning_amd import train_ops as T
    from image_captioning_amd.ops import ACT_SWISH
    M = B * HW
    x, dy = rnd(M, C, seed=1), rnd(M, C, seed=2)
    gate, dpool = torch.sigmoid(rnd(B, C, seed=3)), rnd(B, C, seed=4)
    gamma, beta = rnd(C, seed=5) + 1.5, rnd(C, seed=6)
    eps = 1e-3
    y, saved = T.batchnorm_train(x.to(D), gamma.to(D), beta.to(D), None, None, eps, 0.1, act=ACT_SWISH)
    dx, dg, db = T.batchnorm_act_bwd_se(dy.to(D), x.to(D), saved, gate.to(D), dpool.to(D), 1.0 / HW, HW, act=ACT_SWISH)
    eff = T.chan_affine(dy.to(D), gate.to(D), dpool.to(D), 1.0 / HW, B, HW, C)
    dx2, dg2, db2 = T.batchnorm_act_bwd(eff, x.to(D), saved, act=ACT_SWISH)
    torch.testing.assert_close(dx, dx2, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dg, dg2, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(db, db2, rtol=1e-5, atol=1e-5)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = F.silu(F.batch_norm(xr, None, None, gr, br, True, 0.1, eps))
    eff_r = dy.double().view(B, HW, C) * gate.double()[:, None, :] + dpool.double()[:, None, :] / HW
    yr.backward(eff_r.view(M, C))
    torch.testing.assert_close(dx.cpu().double(), xr.grad, rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(dg.cpu().double(), gr.grad, rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(db.cpu().double(), br.grad, rtol=2e-3, atol=2e-3)


def test_act_pool_and_gated_gradient_argument_errors():
    from image_captioning_amd import train_ops as T
    from image_captioning_amd.ops import ACT_SWISH, IcapError
    x = rnd(6, 6, seed=1).to(D)                                            # 6 channels: not a multiple of 4
    v = rnd(6, seed=2).to(D)
    with pytest.raises(IcapError):
        T.act_pool(x, v, v, ACT_SWISH, 2, 3)
    with pytest.raises(IcapError):
        T.act_pool(rnd(8, 8, seed=1).to(D), rnd(8, seed=2).to(D), rnd(8, seed=2).to(D), ACT_SWISH, 3, 3)     # 8 rows are not 3 x 3


@pytest.mark.parametrize("B,H,W,Cout,K,S", [(2, 224, 224, 32, 3, 2), (3, 17, 23, 32, 3, 2), (2, 16, 16, 24, 3, 1), (5, 9, 9, 8, 1, 1), (1, 225, 225, 32, 3, 2)])
def test_stem_weight_gradient_without_patch_matrix(B, H, W, Cout, K, S):
    """icap_conv_wgrad_gather_f32 against torch autograd (float64) and against the patch-matrix path it replaces."""
    from image_captioning_amd import train_ops as T
    from image_captioning_amd.backbones import same_pad
    lo, hi = same_pad(K, S) if K > 1 else (0, 0)
    x = rnd(B, 3, H, W, seed=1)
    w = rnd(Cout, 3, K, K, seed=2).double().requires_grad_(True)
    y = F.conv2d(F.pad(x.double(), (lo, hi, lo, hi)), w, stride=S)
    Ho, Wo = y.shape[2], y.shape[3]
    dy = rnd(B, Cout, Ho, Wo, seed=3)
    y.backward(dy.double())
    x4 = torch.zeros(B, H, W, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    dz = dy.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout).contiguous()
    dw = T.conv_wgrad_gather(dz.to(D), x4.to(D), B, H, W, 4, 3, K, K, S, lo, lo, Ho, Wo)
    got = dw.view(Cout, K, K, 3).permute(0, 3, 1, 2).cpu().double()
    scale = float(w.grad.abs().max())
    assert float((got - w.grad).abs().max()) <= 2e-5 * scale + 1e-5
    col = T.im2col(x4.to(D), B, H, W, 4, K, K, S, lo, lo, Ho, Wo)
    old = T.gemm_tn(dz.to(D), col).view(Cout, K, K, 4)[..., :3].reshape(Cout, K * K * 3)
    assert float((dw - old).abs().max()) <= 2e-5 * scale + 1e-5
    dw2 = T.conv_wgrad_gather(dz.to(D), x4.to(D), B, H, W, 4, 3, K, K, S, lo, lo, Ho, Wo)
    assert torch.equal(dw, dw2)                                                # fixed summation order


def test_stem_weight_gradient_limits():
    from image_captioning_amd import train_ops as T
    from image_captioning_amd.ops import IcapError
    with pytest.raises(IcapError):                                             # 40 output channels: more than one MFMA block
        T.conv_wgrad_gather(rnd(2 * 4 * 4, 40, seed=1).to(D), rnd(2, 8, 8, 4, seed=2).to(D), 2, 8, 8, 4, 3, 3, 3, 2, 0, 0, 4, 4)
    with pytest.raises(IcapError):                                             # 5 x 5 x 3 = 75 columns
        T.conv_wgrad_gather(rnd(2 * 4 * 4, 8, seed=1).to(D), rnd(2, 8, 8, 4, seed=2).to(D), 2, 8, 8, 4, 3, 5, 5, 2, 1, 1, 4, 4)


def _cnn():
    from image_captioning_amd.backbones import get_cnn_model
    from image_captioning_amd.synthetic import seeded_state_dict
    m = get_cnn_model()
    sd = seeded_state_dict(m, seed=11)
    m.load_state_dict(sd)
    return m, sd


def test_efficientnet_training_forward_backward_matches_oracle():
    B = 3
    model, sd = _cnn()
    model.to(D).train()
    g = torch.Generator().manual_seed(3)
    images = torch.randn(B, 3, 224, 224, generator=g)
    masks = {f"dc{i}": (torch.rand(B, generator=g) >= 0.2 * i / 16).float() / (1 - 0.2 * i / 16) for i in range(16)}
    masks["cnn_drop"] = (torch.rand(B, 1280, generator=g) >= 0.2).float() / 0.8
    dout = torch.randn(B, 512, generator=g)
    # oracle
    sdr = {("cnn_model." + k): v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    bufs = {k: v.detach().clone() for k, v in sdr.items() if "running" in k}
    out_r = BR.cnn_model_train(sdr, images, bufs, masks)
    out_r.backward(dout)
    # HIP
    model.dropout_masks = {k: v.to(D) for k, v in masks.items()}
    out = model(images.to(D))
    out.backward(dout.to(D))
    torch.testing.assert_close(out.detach().cpu(), out_r.detach(), rtol=2e-3, atol=2e-3)
    bad = []
    for name, prm in model.named_parameters():
        want = sdr["cnn_model." + name].grad
        got = prm.grad
        assert got is not None, name
        # a per-channel constant in front of (1x1 conv ->) train-mode BatchNorm is removed by the mean
        # subtraction: _bn2.bias and friends have a true gradient of ~0 and hold rounding noise on both sides,
        # hence the absolute floor
        err = (got.cpu() - want).abs().max().item()
        if err > 2e-2 * want.abs().max().item() + 1e-4:
            bad.append((name, err, want.abs().max().item()))
    assert not bad, bad[:10]
    sd_after = model.state_dict()
    for k in ("0._bn0.running_mean", "0._blocks.5._bn1.running_var", "0._bn1.running_var"):
        torch.testing.assert_close(sd_after[k].cpu(), bufs["cnn_model." + k], rtol=1e-3, atol=1e-5)
