"""GPU: the split-product Linear with pre-split operands streamed straight into LDS (icap_linear_x3dl_f32) - against float64
(fp32 accuracy), against the register-staged split kernel (same six products: results agree to fp32 rounding), its split
result planes against split3 of its fp32 result, and the arg-max partials against a full arg-max."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


SHAPES = [(1024, 512, 2048), (1024, 2048, 512), (1000, 512, 6756), (130, 64, 68), (64, 32, 4), (300, 96, 200), (257, 1024, 132)]


@pytest.mark.parametrize("cfg", [0, 1])
@pytest.mark.parametrize("M,K,N", SHAPES)
def test_linear_x3dl_matches_float64(cfg, M, K, N):
    from image_captioning_amd import ops
    g = torch.Generator().manual_seed(M * 3 + K + N + cfg)
    x = torch.randn(M, K, generator=g).to(dev())
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev())
    bias = torch.randn(N, generator=g).to(dev())
    scale = (torch.rand(N, generator=g) + 0.5).to(dev())
    res = torch.randn(M, N, generator=g).to(dev())
    x3, w3 = ops.split3(x), ops.split3(w)
    y, y3 = ops.linear_x3dl(x3, w3, bias, scale, act=ops.ACT_RELU, residual=res, want_split=True, cfg=cfg)
    ref = torch.relu((x.cpu().double() @ w.cpu().double().t()) * scale.cpu().double() + bias.cpu().double() + res.cpu().double())
    err = float((y.cpu().double() - ref).abs().max())
    assert err <= 2e-6 * max(1.0, float(ref.abs().max())), f"max err {err:.3e}"
    # the register-staged split kernel multiplies the same planes
    y_old = ops.linear_f32x3(x, w3, bias, scale, act=ops.ACT_RELU, residual=res)
    assert float((y - y_old).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    # the split result is the split of the fp32 result, bit for bit
    assert torch.equal(y3.view(torch.int16), ops.split3(y).view(torch.int16))


def test_linear_x3dl_split_only_and_chain():
    """fc1 -> fc2 of the decoder: the first GEMM hands its result over as planes, no fp32 tensor in between."""
    from image_captioning_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(512, 512, generator=g).to(dev())
    w1 = (torch.randn(2048, 512, generator=g) / 512 ** 0.5).to(dev())
    w2 = (torch.randn(512, 2048, generator=g) / 2048 ** 0.5).to(dev())
    b1, b2 = torch.randn(2048, generator=g).to(dev()), torch.randn(512, generator=g).to(dev())
    none, h3 = ops.linear_x3dl(ops.split3(x), ops.split3(w1), b1, act=ops.ACT_RELU, want_f32=False, want_split=True)
    assert none is None
    f, _ = ops.linear_x3dl(h3, ops.split3(w2), b2, residual=x)
    h = torch.relu(x.cpu().double() @ w1.cpu().double().t() + b1.cpu().double())
    ref = h @ w2.cpu().double().t() + b2.cpu().double() + x.cpu().double()
    assert float((f.cpu().double() - ref).abs().max()) <= 5e-6 * float(ref.abs().max())


@pytest.mark.parametrize("M,K,N", [(1024, 512, 6756), (37, 512, 6756), (200, 64, 300), (64, 32, 128)])
def test_linear_x3dl_argmax(M, K, N):
    from image_captioning_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dev())
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev())
    bias = torch.randn(N, generator=g).to(dev())
    pv, pi, logits = ops.linear_x3dl_argmax(ops.split3(x), ops.split3(w), bias, want_logits=True)
    ref = x.cpu().double() @ w.cpu().double().t() + bias.cpu().double()
    assert float((logits.cpu().double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    ids = torch.empty((M, 1), device=dev(), dtype=torch.int64)
    ops.next_token(pv, pi, ids, 0)
    assert torch.equal(ids[:, 0].cpu(), logits.cpu().argmax(dim=1))          # first index on ties, as torch
    pv2, pi2, none = ops.linear_x3dl_argmax(ops.split3(x), ops.split3(w), bias)
    assert none is None and torch.equal(pv, pv2) and torch.equal(pi, pi2)


def test_linear_x3dl_rejects_bad_shapes():
    from image_captioning_amd import ops
    x = torch.randn(64, 40, device=dev())
    w = torch.randn(8, 40, device=dev())
    with pytest.raises(ops.IcapError):
        ops.linear_x3dl(ops.split3(x), ops.split3(w))                             # K % 32
