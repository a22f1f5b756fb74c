"""Training step with fp32-accurate split bf16 products (Trainer(dtype="f32x3")).

Stated tolerance: the forward GEMMs and input gradients of the large layers equal a float64 product as closely as the
fp32 kernel does (error below 1e-5 of the result's largest magnitude - fp32 accumulation over up to 6760 terms; a bf16
product would be 4e-3);
the whole training step's loss equals the fp32 step's to 1e-5 and every non-negligible gradient to 2e-3 in relative L2
norm (the CNN gradients amplify forward differences a few hundred times at this initialisation, see
test_gpu_train_bf16.py; two fp32 summation orders differ by as much)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
D = "cuda:0"


@pytest.mark.parametrize("M,K,N", [(1024, 512, 2048), (5120, 512, 6760), (12544, 1152, 192), (50176, 112, 672)])
def test_linear_and_input_gradient_in_f32x3_step(M, K, N):
    from image_captioning_amd import ops, train_ops as T
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    dy = torch.randn(M, N, generator=g)
    acc = torch.randn(M, K, generator=g)
    xd, wd, bd, dyd = x.to(D), w.to(D), b.to(D), dy.to(D)
    seen = []
    ops.set_trace(lambda n, s: seen.append(n))
    try:
        with ops.train_step("f32x3"):
            y = ops.linear(xd, wd, bd)
            dx = T.gemm_nn(dyd, wd)
            dx2 = T.gemm_nn(dyd, wd, out=acc.to(D), accumulate=True)
    finally:
        ops.set_trace(None)
    want_y = x.double() @ w.double().t() + b.double()
    want_dx = dy.double() @ w.double()
    for got, want in ((y, want_y), (dx, want_dx), (dx2, want_dx + acc.double())):
        err = (got.cpu().double() - want).abs().max().item() / want.abs().max().item()
        assert err < 1e-5, err
    if K >= ops.X3_MIN_K and N >= ops.X3_MIN_K:
        assert "linear_f32x3" in seen, seen                           # these shapes really take the split-product kernel
    assert ops.TRAIN_DTYPE == "f32" and ops._step_cache is None


def test_small_gemms_stay_on_the_fp32_kernel():
    from image_captioning_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(256, 512, generator=g).to(D)
    w = torch.randn(2048, 512, generator=g).to(D)
    with ops.train_step("f32x3"):
        a = ops.linear(x, w)
    assert torch.equal(a, ops.linear(x, w))


def test_full_training_step_f32x3_equals_f32():
    from image_captioning_amd import transformer_word_level_bahdnau as my_bah
    from image_captioning_amd._bah_train import DropoutState
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_captions, synthetic_images
    from image_captioning_amd.training import Trainer
    V, B, T = 300, 32, 6
    images = synthetic_images(B, seed=3).to(D)
    caps = synthetic_captions(B, T, V, 2, 3, 0, seed=4).to(D)
    out = {}
    for kind in ("f32", "f32x3"):
        model = my_bah.build_model(V, dropout=0.3)
        model.load_state_dict(seeded_state_dict(model, seed=5))
        model.to(D)
        DropoutState.manual_seed(77)
        trainer = Trainer(model, lr=1e-4, weight_decay=1e-5, dtype=kind)
        loss = trainer.step(images, caps, 0)
        out[kind] = (loss.item(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    (l32, g32), (lx3, gx3) = out["f32"], out["f32x3"]
    assert abs(lx3 - l32) < 1e-5 * abs(l32), (lx3, l32)
    gmax = max(v.abs().max().item() for v in g32.values())
    bad, checked = [], 0
    for n, want in g32.items():
        if want.abs().max().item() < 1e-3 * gmax:
            continue      # e.g. a bias in front of a train-mode BatchNorm: its exact gradient is 0, both steps hold rounding noise
        rel = ((gx3[n] - want).norm() / want.norm()).item()
        checked += 1
        if rel > 2e-3:
            bad.append((n, rel))
    assert checked > 100 and not bad, (checked, bad[:10])
