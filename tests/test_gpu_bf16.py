"""GPU: the opt-in bf16 compute mode (bf16 products, fp32 accumulation, fp32 tensors) against the fp32 mode, which
tests/test_gpu_models.py pins to the oracle.  Stated bf16 tolerance: encoder features within 1e-2 of the largest
feature magnitude, logits within 2e-2 of the largest logit magnitude, greedy token ids identical wherever the fp32
top-2 logit gap exceeds twice the observed logit difference."""
import pytest
import torch

pytestmark = pytest.mark.gpu
D = "cuda:0"


@pytest.fixture
def bf16_mode():
    from image_captioning_amd import ops
    yield ops
    ops.set_compute_dtype("f32")


def test_bf16_mode_tracks_the_fp32_path(bf16_mode):
    ops = bf16_mode
    from image_captioning_amd import transformer_word_level as twl
    from image_captioning_amd.synthetic import seeded_state_dict
    model = twl.build_model(6756)
    model.load_state_dict(seeded_state_dict(model, seed=5))
    model = model.to(D).eval()
    g = torch.Generator().manual_seed(11)
    imgs = torch.randn(12, 3, 224, 224, generator=g).to(D)          # B > 8: the batched GEMM decode
    res = {}
    for mode in ("f32", "bf16"):
        ops.set_compute_dtype(mode)
        feats = model.cnn_model(imgs)
        ids, logits = model.greedy_ids_from_features(feats, 6, want_logits=True)
        res[mode] = (feats.clone(), ids.clone(), logits.clone())
    f0, i0, l0 = res["f32"]
    f1, i1, l1 = res["bf16"]
    assert not torch.equal(f0, f1)                                    # the bf16 kernels really ran
    assert float((f0 - f1).abs().max()) <= 1e-2 * float(f0.abs().max())
    # step 0 sees the same inputs in both modes up to the feature difference
    d0 = float((l0[0] - l1[0]).abs().max())
    assert d0 <= 2e-2 * float(l0[0].abs().max())
    top2 = torch.topk(l0[0], 2, dim=-1).values
    decided = (top2[:, 0] - top2[:, 1]) > 2 * d0
    assert torch.equal(i0[decided, 0], i1[decided, 0])


BF16_FEATURE_TOL = 1e-2      # of the largest feature magnitude, against the fp32 CPU oracle
BF16_LOGIT_TOL = 2e-2        # of the largest logit magnitude


@pytest.mark.parametrize("cnn_type", ["resnet18", "vgg16"])
def test_bf16_dense_encoders_against_the_cpu_oracle(bf16_mode, cnn_type):
    """ResNet-18 / VGG-16 with bf16 products and bf16 activations between layers, against the fp32 CPU restatement of the
    encoder (not against this library's own fp32 path): features within the stated bf16 tolerance."""
    ops = bf16_mode
    from image_captioning_amd import seq2seqbasic as s2s
    from image_captioning_amd.preprocess import to_model_input
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_images
    from oracle import backbones_ref as BR
    from tests import helpers as H
    enc = s2s.EncoderCNN(256, cnn_type)
    enc.load_state_dict(seeded_state_dict(enc, seed=41))
    enc.eval()
    sd = {"encoder." + k: v for k, v in H.cpu_sd(enc).items()}
    imgs = torch.cat([to_model_input(H.load("food_test_u8")["images"])[:3], synthetic_images(1)])
    want = BR.seq2seq_encoder(sd, imgs, cnn_type)
    ops.set_compute_dtype("bf16")
    got = enc.to(D)(imgs.to(D)).cpu()
    err = float((got - want).abs().max())
    assert 0 < err <= BF16_FEATURE_TOL * float(want.abs().max()), (err, float(want.abs().max()))


def test_bf16_captioner_against_the_cpu_oracle(bf16_mode):
    """EfficientNet-B0 + word-level transformer in bf16 mode on the food photos against the CPU oracle: features and
    first-step logits within the stated tolerance, greedy ids equal wherever the ORACLE's top-2 gap exceeds twice the
    observed logit error (a bf16 logit error can legitimately flip closer calls)."""
    ops = bf16_mode
    from image_captioning_amd import transformer_word_level as twl
    from image_captioning_amd.preprocess import to_model_input
    from image_captioning_amd.synthetic import seeded_state_dict
    from oracle import backbones_ref as BR, torch_ref as R
    from tests import helpers as H
    V, steps = 6756, 6
    model = twl.build_model(V)
    model.load_state_dict(seeded_state_dict(model, seed=5))
    model.eval()
    sd = H.cpu_sd(model)
    imgs = to_model_input(H.load("food_test_u8")["images"])[:4]
    with torch.no_grad():
        f_ref = BR.cnn_model(sd, imgs)
        runs = [R.tfm_greedy(sd, f, 8, 8, steps, 2, 3, False) for f in f_ref]
    ops.set_compute_dtype("bf16")
    model = model.to(D)
    feats = model.cnn_model(imgs.to(D))
    ids, logits = model.greedy_ids_from_features(feats, steps, want_logits=True, lockstep=0)     # the batched GEMM path (bf16 GEMMs)
    ferr = float((feats.cpu() - f_ref).abs().max())
    assert 0 < ferr <= BF16_FEATURE_TOL * float(f_ref.abs().max())
    for r, (want_ids, want_logits) in enumerate(runs):
        lerr = float((logits[0, r].cpu() - want_logits[0]).abs().max())
        assert lerr <= BF16_LOGIT_TOL * float(want_logits[0].abs().max()), (r, lerr)
        top2 = torch.topk(want_logits, 2, dim=1).values
        gaps = (top2[:, 0] - top2[:, 1]).tolist()
        for t in range(steps):
            if gaps[t] <= 2 * max(lerr, 1e-6) * 4:       # later steps accumulate nothing (no recurrence), but keep a margin
                break
            assert int(ids[r, t]) == want_ids[t], (r, t, gaps[t], lerr)


def test_compute_dtype_switch_and_fallback_shapes(bf16_mode):
    ops = bf16_mode
    with pytest.raises(ops.IcapError):
        ops.set_compute_dtype("fp8")
    ops.set_compute_dtype("bf16")
    # a 1x1 convolution whose channel counts the bf16 kernel does not take (Cin % 8) stays on the fp32 kernel
    x = torch.randn(2, 5, 5, 12, device=D)
    w = torch.randn(20, 1, 1, 12, device=D)
    cache = {}
    y = ops.pointwise_conv(x, w, cache, "k")
    assert not cache and torch.allclose(y, torch.einsum("bhwc,oc->bhwo", x, w.view(20, 12)), rtol=1e-4, atol=1e-4)
    x = torch.randn(2, 5, 5, 16, device=D)
    w = torch.randn(20, 1, 1, 16, device=D)
    y = ops.pointwise_conv(x, w, cache, "k")
    assert cache["k"].dtype == torch.bfloat16
    ref = torch.einsum("bhwc,oc->bhwo", x.to(torch.bfloat16).float(), w.view(20, 16).to(torch.bfloat16).float())
    assert torch.allclose(y, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("Cin,Cexp,H,W,ks,s,pad", [(16, 96, 37, 41, 3, 2, (0, 1, 0, 1)), (24, 144, 28, 30, 3, 1, (1, 1, 1, 1)),
                                                    (24, 144, 29, 27, 5, 2, (1, 2, 1, 2)), (40, 240, 28, 17, 5, 1, (2, 2, 2, 2)),
                                                    (40, 240, 28, 28, 3, 2, (0, 1, 0, 1))])
def test_fused_expand_dwconv_bf16(bf16_mode, Cin, Cexp, H, W, ks, s, pad):
    """The fused kernel with bf16 products in its expand GEMM: against torch on bf16-rounded expand operands (the BN scale
    is folded into the weights before they are rounded, so the comparison allows a few bf16 ulps of the expanded value)."""
    import torch.nn.functional as F
    ops = bf16_mode
    g = lambda seed: torch.Generator().manual_seed(seed)
    B = 2
    x = torch.randn(B, Cin, H, W, generator=g(1))
    we = torch.randn(Cexp, Cin, 1, 1, generator=g(2)) * Cin ** -0.5
    es, eb = torch.rand(Cexp, generator=g(3)) + 0.5, torch.randn(Cexp, generator=g(4)) * 0.3
    wd = torch.randn(Cexp, 1, ks, ks, generator=g(5)) / ks
    dsc, dbi = torch.rand(Cexp, generator=g(6)) + 0.5, torch.randn(Cexp, generator=g(7)) * 0.1
    e = F.conv2d(x, we) * es[None, :, None, None] + eb[None, :, None, None]
    e = e * torch.sigmoid(e)
    ref = F.conv2d(F.pad(e, (pad[2], pad[3], pad[0], pad[1])), wd, stride=s, groups=Cexp) * dsc[None, :, None, None] + dbi[None, :, None, None]
    ref = ref * torch.sigmoid(ref)
    args = (x.permute(0, 2, 3, 1).contiguous().to(D), we.view(Cexp, Cin).contiguous().to(D), es.to(D), eb.to(D),
            wd.view(Cexp, ks * ks).t().contiguous().to(D), dsc.to(D), dbi.to(D))
    ops.set_compute_dtype("f32")
    y32, _ = ops.expand_dwconv2d(*args, ks=ks, stride=s, pad=pad, want_partial=True)
    ops.set_compute_dtype("bf16")
    y16, part = ops.expand_dwconv2d(*args, ks=ks, stride=s, pad=pad, want_partial=True)
    assert not torch.equal(y16, y32)
    scale = float(ref.abs().max())
    assert float((y16.permute(0, 3, 1, 2).cpu() - ref).abs().max()) <= 2e-2 * scale      # stated bf16 tolerance
    assert float((y16 - y32).abs().max()) <= 2e-2 * scale
    torch.testing.assert_close(part.sum(1).cpu() / (ref.shape[2] * ref.shape[3]), y16.mean((1, 2)).cpu(), rtol=1e-4, atol=1e-5)
    # bf16 result tensor: the same values rounded to nearest even on the store; the SE partial sums stay those of the fp32 values
    yb, partb = ops.expand_dwconv2d(*args, ks=ks, stride=s, pad=pad, want_partial=True, out_bf16=True)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb, y16.to(torch.bfloat16)) and torch.equal(partb, part)
    ops.set_compute_dtype("f32")
    with pytest.raises(ops.IcapError):
        ops.expand_dwconv2d(*args, ks=ks, stride=s, pad=pad, out_bf16=True)


def test_fused_stem_dwconv_bf16_result(bf16_mode):
    ops = bf16_mode
    g = lambda seed: torch.Generator().manual_seed(seed)
    x = torch.randn(2, 45, 51, 4, generator=g(1)).to(D)
    x[..., 3] = 0
    ws = (torch.randn(32, 3, 3, 4, generator=g(2)) * 0.2).to(D)
    ws[..., 3] = 0
    ss, sb = (torch.rand(32, generator=g(3)) + 0.5).to(D), (torch.randn(32, generator=g(4)) * 0.3).to(D)
    wd = (torch.randn(9, 32, generator=g(5)) / 3).to(D)
    dsc, dbi = (torch.rand(32, generator=g(6)) + 0.5).to(D), (torch.randn(32, generator=g(7)) * 0.1).to(D)
    ops.set_compute_dtype("bf16")
    y, part = ops.stem_dwconv2d(x, ws, ss, sb, wd, dsc, dbi, want_partial=True)
    yb, partb = ops.stem_dwconv2d(x, ws, ss, sb, wd, dsc, dbi, want_partial=True, out_bf16=True)
    assert yb.dtype == torch.bfloat16 and torch.equal(yb, y.to(torch.bfloat16)) and torch.equal(partb, part)


@pytest.mark.parametrize("M,K,N,hw", [(2 * 49, 96, 24, 49), (300, 240, 40, 100), (37, 32, 16, 37), (513, 144, 132, 171)])
def test_linear_bf16_takes_bf16_activations(bf16_mode, M, K, N, hw):
    """A bf16 activation tensor gives exactly what its fp32 copy gives without a gate (bf16 -> fp32 -> bf16 is the identity);
    with an SE gate the product gate*x is rounded once more, so it is checked against torch on that rounded product."""
    ops = bf16_mode
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(D).to(torch.bfloat16)
    w16 = ops.to_bf16((torch.randn(N, K, generator=g) * K ** -0.5).to(D))
    b, sc = torch.randn(N, generator=g).to(D), (torch.rand(N, generator=g) + 0.5).to(D)
    res = torch.randn(M, N, generator=g).to(D)
    y16 = ops.linear_bf16(x, w16, b, sc, residual=res, act=ops.ACT_SWISH)
    y32 = ops.linear_bf16(x.float(), w16, b, sc, residual=res, act=ops.ACT_SWISH)
    assert torch.equal(y16, y32)
    gate = torch.rand(M // hw, K, generator=g).to(D)
    yg = ops.linear_bf16(x, w16, b, sc, gate=gate, hw=hw)
    xg = (x.float().view(M // hw, hw, K) * gate[:, None, :]).to(torch.bfloat16).float().view(M, K)
    ref = (xg.double() @ w16.float().double().t()) * sc.double() + b.double()
    torch.testing.assert_close(yg, ref.float(), rtol=1e-4, atol=1e-4)
    with pytest.raises(ops.IcapError):                                  # the fp32 kernels do not take bf16 tensors
        ops.conv2d(x.view(1, 1, M, K), torch.randn(N, 1, 1, K, device=D))


@pytest.mark.parametrize("M,K,N", [(300, 512, 6756), (7, 64, 84), (1024, 512, 100)])
def test_linear_bf16_argmax_partials(bf16_mode, M, K, N):
    """The arg-max partials of the bf16 vocabulary projection reduce to the row arg-max of its own logits."""
    ops = bf16_mode
    g = torch.Generator().manual_seed(M + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g)
    w16 = ops.to_bf16(w.to(D))
    pv, pi, logits = ops.linear_bf16_argmax(x.to(D), w16, b.to(D), want_logits=True)
    ref = ops.linear_bf16(x.to(D), w16, b.to(D))
    torch.testing.assert_close(logits, ref, rtol=0, atol=0)
    best = pv.argmax(1)
    ids = pi.gather(1, best[:, None]).squeeze(1).long()
    assert torch.equal(ids, ref.argmax(1))
    assert torch.equal(pv.max(1).values, ref.max(1).values)
    pv2, pi2, none = ops.linear_bf16_argmax(x.to(D), w16, b.to(D))
    assert none is None and torch.equal(pv2, pv) and torch.equal(pi2, pi)


@pytest.mark.parametrize("B,H,W,Cin,Cout,K,stride,pad,res", [(2, 14, 14, 64, 64, 3, 1, (1, 1, 1, 1), True), (3, 9, 11, 32, 132, 3, 2, (1, 1, 1, 1), False),
                                                             (2, 28, 28, 128, 256, 1, 2, (0, 0, 0, 0), False), (1, 7, 7, 512, 512, 3, 1, (1, 1, 1, 1), True),
                                                             (5, 20, 16, 96, 40, 5, 1, (2, 2, 2, 2), False),
                                                             (2, 37, 41, 4, 64, 7, 2, (3, 3, 3, 3), False), (2, 18, 18, 4, 36, 3, 1, (1, 1, 1, 1), False)])
def test_dense_conv_bf16(bf16_mode, B, H, W, Cin, Cout, K, stride, pad, res):
    """The bf16 implicit-GEMM convolution against torch on bf16-rounded operands (exact products, fp32 accumulation)."""
    import torch.nn.functional as F
    ops = bf16_mode
    g = torch.Generator().manual_seed(B * 100 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, K, K, generator=g) * (Cin * K * K) ** -0.5
    sc, bi = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    xb, wb = x.to(torch.bfloat16).double(), w.to(torch.bfloat16).double()
    ref = F.conv2d(F.pad(xb, (pad[2], pad[3], pad[0], pad[1])), wb, stride=stride) * sc.double()[None, :, None, None] + bi.double()[None, :, None, None]
    r = torch.randn(ref.shape, generator=g) if res else None
    if res:
        ref = ref + r.double()
    ref = torch.relu(ref).float()
    xd = x.permute(0, 2, 3, 1).contiguous().to(D)
    wd = w.permute(0, 2, 3, 1).contiguous().to(D)
    rd = r.permute(0, 2, 3, 1).contiguous().to(D) if res else None
    cache = {}
    ops.set_compute_dtype("bf16")
    y = ops.dense_conv(xd, wd, cache, "k", sc.to(D), bi.to(D), residual=rd, stride=stride, pad=pad, act=ops.ACT_RELU)
    assert cache["k"].dtype == torch.bfloat16
    torch.testing.assert_close(y.permute(0, 3, 1, 2).cpu(), ref, rtol=1e-4, atol=1e-4)
    ops.set_compute_dtype("f32")
    y32 = ops.dense_conv(xd, wd, {}, "k", sc.to(D), bi.to(D), residual=rd, stride=stride, pad=pad, act=ops.ACT_RELU)
    assert float((y - y32).abs().max()) <= 2e-2 * float(y32.abs().max())           # stated bf16 tolerance vs the fp32 kernel
    ops.set_compute_dtype("bf16")
    x12 = torch.randn(1, 5, 5, 12, device=D)                                         # Cin % 32 != 0 stays on the fp32 kernel
    w12 = torch.randn(8, 3, 3, 12, device=D)
    c2 = {}
    ops.dense_conv(x12, w12, c2, "k", pad=(1, 1, 1, 1))
    assert not c2


@pytest.mark.parametrize("kind", ["resnet18", "vgg16"])
def test_dense_encoders_bf16_mode_track_fp32(bf16_mode, kind):
    ops = bf16_mode
    from image_captioning_amd import seq2seqbasic as s2s
    from image_captioning_amd.synthetic import seeded_state_dict
    enc = s2s.EncoderCNN(256, cnn_type=kind)
    enc.load_state_dict(seeded_state_dict(enc, seed=9))
    enc = enc.to(D).eval()
    imgs = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(4)).to(D)
    ops.set_compute_dtype("f32")
    f32 = enc(imgs).clone()
    ops.set_compute_dtype("bf16")
    f16 = enc(imgs).clone()
    assert not torch.equal(f32, f16)
    assert float((f32 - f16).abs().max()) <= 1e-2 * float(f32.abs().max())


@pytest.mark.parametrize("B,H,W,Cin,Cout,K,stride,res", [(2, 14, 14, 64, 64, 3, 1, True), (3, 9, 11, 32, 160, 3, 2, False),
                                                         (1, 7, 7, 512, 512, 3, 1, True), (2, 28, 28, 128, 256, 1, 2, False)])
def test_dense_conv_bf16_storage(bf16_mode, B, H, W, Cin, Cout, K, stride, res):
    """bf16 input / residual / result tensors: same arithmetic as with fp32 tensors holding the same (bf16-representable)
    values, the result rounded to nearest even on the store."""
    ops = bf16_mode
    ops.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(B * 10 + Cin)
    pad = (K // 2,) * 4
    x16 = torch.randn(B, H, W, Cin, generator=g).to(D).to(torch.bfloat16)
    w = (torch.randn(Cout, K, K, Cin, generator=g) * (Cin * K * K) ** -0.5).to(D)
    sc, bi = (torch.rand(Cout, generator=g) + 0.5).to(D), (torch.randn(Cout, generator=g) * 0.1).to(D)
    Ho, Wo = ops.conv_out_size(H, K, stride, pad[0], pad[1]), ops.conv_out_size(W, K, stride, pad[2], pad[3])
    r16 = torch.randn(B, Ho, Wo, Cout, generator=g).to(D).to(torch.bfloat16) if res else None
    cache = {}
    ref = ops.dense_conv(x16.float(), w, cache, "k", sc, bi, residual=r16.float() if res else None, stride=stride, pad=pad, act=ops.ACT_RELU)
    y_f = ops.dense_conv(x16, w, cache, "k", sc, bi, residual=r16, stride=stride, pad=pad, act=ops.ACT_RELU)
    assert y_f.dtype == torch.float32 and torch.equal(y_f, ref)
    y_h = ops.dense_conv(x16, w, cache, "k", sc, bi, residual=r16, stride=stride, pad=pad, act=ops.ACT_RELU, out_bf16=True)
    assert y_h.dtype == torch.bfloat16 and torch.equal(y_h, ref.to(torch.bfloat16))
    y_m = ops.dense_conv(x16.float(), w, cache, "k", sc, bi, residual=r16, stride=stride, pad=pad, act=ops.ACT_RELU, out_bf16=True)
    assert torch.equal(y_m, y_h)
    ops.set_compute_dtype("f32")
    with pytest.raises(ops.IcapError):
        ops.dense_conv(x16, w, {}, "k", sc, bi, stride=stride, pad=pad)


@pytest.mark.parametrize("B,H,W,Cin,Cout,K,stride,res", [
    (2, 14, 14, 64, 64, 3, 1, True), (3, 9, 11, 64, 160, 3, 2, False), (1, 7, 7, 512, 512, 3, 1, True), (2, 28, 28, 128, 256, 1, 2, False),
    (5, 16, 16, 128, 128, 3, 1, False), (1, 30, 23, 64, 72, 3, 1, True), (2, 12, 12, 192, 264, 3, 1, False), (1, 3, 3, 64, 8, 3, 1, False)])
def test_dense_conv_bf16_wide(bf16_mode, B, H, W, Cin, Cout, K, stride, res):
    """cfg 5 of icap_conv2d_bf16_f32 (256-row tiles, eight waves, direct-to-LDS loads, bf16 in and out) against the tile kernel on
    the same bf16 tensors: every tile width, ragged last row / column tile, stride 2, 1x1, residual; sums in another order, so
    the bf16 results may differ by one rounding of values whose fp32 sums differ in the last bits."""
    ops = bf16_mode
    ops.set_compute_dtype("bf16")
    g = torch.Generator().manual_seed(B * 10 + Cin + Cout)
    pad = (K // 2,) * 4
    x16 = torch.randn(B, H, W, Cin, generator=g).to(D).to(torch.bfloat16)
    w = (torch.randn(Cout, K, K, Cin, generator=g) * (Cin * K * K) ** -0.5).to(D)
    sc, bi = (torch.rand(Cout, generator=g) + 0.5).to(D), (torch.randn(Cout, generator=g) * 0.1).to(D)
    Ho, Wo = ops.conv_out_size(H, K, stride, pad[0], pad[1]), ops.conv_out_size(W, K, stride, pad[2], pad[3])
    r16 = torch.randn(B, Ho, Wo, Cout, generator=g).to(D).to(torch.bfloat16) if res else None
    cache = {}
    for act in (ops.ACT_RELU, ops.ACT_NONE):
        ref32 = ops.dense_conv(x16, w, cache, "k", sc, bi, residual=r16, stride=stride, pad=pad, act=act, cfg=2)
        y = ops.dense_conv(x16, w, cache, "k", sc, bi, residual=r16, stride=stride, pad=pad, act=act, out_bf16=True, cfg=5)
        assert y.dtype == torch.bfloat16 and tuple(y.shape) == (B, Ho, Wo, Cout)
        err = (y.float() - ref32).abs()
        tol = 2.0 ** -8 * ref32.abs() + 1e-5 * float(ref32.abs().max())       # one bf16 rounding (half an ulp = 2^-9 relative) and change
        assert bool((err <= tol).all()), float((err - tol).max())
        assert float((y.float() - ref32.to(torch.bfloat16).float()).abs().max()) <= 2.0 ** -7 * float(ref32.abs().max())
    with pytest.raises(ops.IcapError):       # fp32 tensors, or channels that are not whole K tiles, are an error for this configuration
        ops.dense_conv(x16.float(), w, cache, "k", sc, bi, stride=stride, pad=pad, cfg=5)


def test_pooling_of_bf16_activations(bf16_mode):
    import torch.nn.functional as F
    ops = bf16_mode
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3, 13, 17, 64, generator=g).to(D).to(torch.bfloat16)
    for ks, st, pad in ((2, 2, 0), (3, 2, 1)):
        y = ops.maxpool2d(x, ks, st, pad)
        ref = F.max_pool2d(x.float().permute(0, 3, 1, 2), ks, st, pad).permute(0, 2, 3, 1)
        assert y.dtype == torch.bfloat16 and torch.equal(y.float(), ref)
    a = ops.adaptive_avgpool2d(x, 1, 1)
    assert a.dtype == torch.float32
    torch.testing.assert_close(a.view(3, 64), x.float().mean((1, 2)), rtol=1e-5, atol=1e-6)
    x7 = torch.randn(2, 7, 7, 32, generator=g).to(D).to(torch.bfloat16)
    assert torch.equal(ops.adaptive_avgpool2d(x7, 7, 7), x7.float())                  # the bf16 -> fp32 step of the VGG trunk
    a2 = ops.adaptive_avgpool2d(x, 7, 7)
    ref2 = F.adaptive_avg_pool2d(x.float().permute(0, 3, 1, 2), (7, 7)).permute(0, 2, 3, 1)
    torch.testing.assert_close(a2, ref2, rtol=1e-5, atol=1e-6)


def test_f32x3_mode_matches_the_cpu_oracle_like_fp32(bf16_mode):
    """"f32x3": fp32-accurate products on the bf16 matrix cores.  Unlike bf16 mode it must meet the FP32 bar: encoder
    features and logits within rtol 1e-3 of the CPU oracle, greedy ids exact under the 1e-4 gap rule - with and without
    the per-shape tuner choosing between the split kernel's configurations and the fp32 kernel."""
    ops = bf16_mode
    from image_captioning_amd import transformer_word_level as twl
    from image_captioning_amd.preprocess import to_model_input
    from image_captioning_amd.synthetic import seeded_state_dict, synthetic_images
    from oracle import backbones_ref as BR, torch_ref as R
    from tests import helpers as H
    V, steps, n_or = 6756, 6, 4
    model = twl.build_model(V)
    model.load_state_dict(seeded_state_dict(model, seed=5))
    model.eval()
    sd = H.cpu_sd(model)
    imgs = torch.cat([to_model_input(H.load("food_test_u8")["images"]), synthetic_images(24, seed=3)])     # 32 images
    imgs = imgs.repeat(8, 1, 1, 1)                                                                           # 256: the GEMM path's split kernels engage
    with torch.no_grad():
        f_ref = BR.cnn_model(sd, imgs[:n_or])
        runs = [R.tfm_greedy(sd, f, 8, 8, steps, 2, 3, False) for f in f_ref]
    model = model.to(D)
    x = imgs.to(D)
    ops.set_compute_dtype("f32")
    f32_feats = model.cnn_model(x).clone()
    for tune in (False, True):
        ops.set_compute_dtype("f32x3")
        ops.set_autotune(tune)
        try:
            feats = model.cnn_model(x)
            ids, logits = model.greedy_ids_from_features(feats, steps, want_logits=True)
        finally:
            ops.set_autotune(False)
        assert not torch.equal(feats, f32_feats), "the split kernels did not run"
        err = (feats[:n_or].cpu() - f_ref).abs()
        assert float((err - (1e-3 * f_ref.abs() + 1e-3 * float(f_ref.abs().max()))).max()) <= 0
        assert float((feats - f32_feats).abs().max()) <= 2e-5 * float(f32_feats.abs().max())      # a few fp32 roundings per layer
        for r, (want_ids, want_logits) in enumerate(runs):
            got = logits[:, r].cpu()
            assert float(((got - want_logits).abs() - (1e-4 + 1e-3 * want_logits.abs())).max()) <= 0
            top2 = torch.topk(want_logits, 2, dim=1).values
            gaps = (top2[:, 0] - top2[:, 1]).unsqueeze(0).numpy()
            H.assert_ids_match(ids[r:r + 1].cpu().numpy(), [want_ids], gaps, f"f32x3 image {r} tune={tune}")
        assert torch.equal(ids[:32], ids[32:64])          # the tiled copies of an image decode alike
