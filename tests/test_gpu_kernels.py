"""GPU: every C-ABI kernel against a plain fp32 torch CPU computation of the same op."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 1e-3   # fp32 tolerance of the hot path (features / logits)


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(got, want, rtol=RTOL, atol=1e-4, what=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, f"{what}: {tuple(got.shape)} vs {tuple(want.shape)}"
    err = (got - want).abs() - (atol + rtol * want.abs())
    assert err.max() <= 0, f"{what}: max abs diff {(got - want).abs().max():.3e} (max |want| {want.abs().max():.3e})"


def act_ref(x, act):
    from image_captioning_amd import ops
    return {ops.ACT_NONE: lambda v: v, ops.ACT_RELU: torch.relu, ops.ACT_SWISH: lambda v: v * torch.sigmoid(v),
            ops.ACT_SIGMOID: torch.sigmoid, ops.ACT_TANH: torch.tanh}[act](x)


CONV_CASES = [
    # B, H, W, Cin, Cout, K, stride, pad(t,b,l,r), act, residual, gate
    (2, 16, 16, 8, 32, 3, 1, (1, 1, 1, 1), 1, False, False),
    (3, 17, 15, 16, 40, 3, 2, (0, 1, 0, 1), 2, False, False),       # TF-same asymmetric, odd sizes
    (2, 23, 23, 4, 64, 7, 2, (3, 3, 3, 3), 1, False, False),        # resnet stem shape, padded 3->4 channels
    (2, 14, 14, 24, 144, 1, 1, (0, 0, 0, 0), 2, False, False),      # K=24: tail of the 16-wide K slice masked
    (2, 14, 14, 144, 24, 1, 1, (0, 0, 0, 0), 0, True, True),        # MBConv project: SE gate + skip
    (2, 14, 14, 64, 128, 1, 2, (0, 0, 0, 0), 0, False, False),      # resnet downsample
    (1, 56, 56, 64, 64, 3, 1, (1, 1, 1, 1), 1, True, False),        # basic block tail: residual then relu
    (5, 9, 9, 32, 200, 3, 1, (1, 1, 1, 1), 0, False, False),        # Cout not a multiple of the tile
    (40, 28, 28, 40, 96, 1, 1, (0, 0, 0, 0), 2, False, False),      # many rows, 128x128 tile path
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(case):
    from image_captioning_amd import ops
    B, H, W, Cin, Cout, K, s, pad, act, use_res, use_gate = case
    d = dev()
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, K, K, seed=2, scale=(Cin * K * K) ** -0.5)
    scale = torch.rand(Cout, generator=torch.Generator().manual_seed(3)) + 0.5
    bias = rnd(Cout, seed=4, scale=0.1)
    gate = torch.rand(B, Cin, generator=torch.Generator().manual_seed(5)) if use_gate else None
    xin = x * gate[:, :, None, None] if use_gate else x
    ref = F.conv2d(F.pad(xin, (pad[2], pad[3], pad[0], pad[1])), w, stride=s)
    ref = ref * scale[None, :, None, None] + bias[None, :, None, None]
    res = rnd(*ref.shape, seed=6) if use_res else None
    if use_res:
        ref = ref + res
    ref = act_ref(ref, act)
    y = ops.conv2d(x.permute(0, 2, 3, 1).contiguous().to(d), w.permute(0, 2, 3, 1).contiguous().to(d), scale.to(d), bias.to(d),
                   residual=res.permute(0, 2, 3, 1).contiguous().to(d) if use_res else None,
                   gate=gate.to(d) if use_gate else None, stride=s, pad=pad, act=act)
    close(y.permute(0, 3, 1, 2), ref, what=f"conv {case}")


@pytest.mark.parametrize("M,K,N", [(1, 512, 256), (8, 256, 512), (120, 512, 2048), (256, 2048, 512), (300, 512, 6756), (2000, 64, 97)])
def test_linear(M, K, N):
    from image_captioning_amd import ops
    d = dev()
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    y = ops.linear(x.to(d), w.to(d), b.to(d), act=ops.ACT_RELU)
    close(y, torch.relu(x @ w.t() + b), what=f"linear {M}x{K}x{N}")


def test_nchw_to_nhwc_pad():
    from image_captioning_amd import ops
    x = rnd(3, 3, 10, 7, seed=1)
    y = ops.nchw_to_nhwc_pad(x.to(dev()), 4).cpu()
    assert torch.equal(y[..., :3], x.permute(0, 2, 3, 1))
    assert torch.equal(y[..., 3], torch.zeros(3, 10, 7))


@pytest.mark.parametrize("C,H,W,ks,s,pad", [(32, 20, 20, 3, 1, (1, 1, 1, 1)), (96, 21, 19, 3, 2, (0, 1, 0, 1)),
                                             (144, 14, 14, 5, 2, (1, 2, 1, 2)), (240, 9, 9, 5, 1, (2, 2, 2, 2)),
                                             (1152, 7, 7, 3, 1, (1, 1, 1, 1)), (672, 14, 14, 5, 1, (2, 2, 2, 2))])
def test_dwconv_and_se(C, H, W, ks, s, pad):
    from image_captioning_amd import ops
    d = dev()
    B = 3
    x = rnd(B, C, H, W, seed=1)
    w = rnd(C, 1, ks, ks, seed=2, scale=1.0 / ks)
    scale = torch.rand(C, generator=torch.Generator().manual_seed(3)) + 0.5
    bias = rnd(C, seed=4, scale=0.1)
    ref = F.conv2d(F.pad(x, (pad[2], pad[3], pad[0], pad[1])), w, stride=s, groups=C)
    ref = ref * scale[None, :, None, None] + bias[None, :, None, None]
    ref = ref * torch.sigmoid(ref)
    y, part = ops.dwconv2d(x.permute(0, 2, 3, 1).contiguous().to(d), w.view(C, ks * ks).t().contiguous().to(d), scale.to(d),
                           bias.to(d), ks=ks, stride=s, pad=pad, act=ops.ACT_SWISH, want_partial=True)
    close(y.permute(0, 3, 1, 2), ref, what="dwconv")
    hw = ref.shape[2] * ref.shape[3]
    close(part.sum(1) / hw, ref.mean((2, 3)), what="dwconv partial pool")
    S = max(1, C // 24)
    wr, br, we, be = rnd(S, C, seed=5, scale=C ** -0.5), rnd(S, seed=6), rnd(C, S, seed=7, scale=S ** -0.5), rnd(C, seed=8)
    gate = ops.se_gate(part, hw, wr.to(d), br.to(d), we.t().contiguous().to(d), be.to(d))
    sq = ref.mean((2, 3)) @ wr.t() + br
    sq = sq * torch.sigmoid(sq)
    close(gate, torch.sigmoid(sq @ we.t() + be), what="se gate")


@pytest.mark.parametrize("Cin,Cexp,H,W,ks,s,pad", [(16, 96, 37, 41, 3, 2, (0, 1, 0, 1)), (24, 144, 28, 30, 3, 1, (1, 1, 1, 1)),
                                                    (24, 144, 29, 27, 5, 2, (1, 2, 1, 2)), (40, 240, 28, 17, 5, 1, (2, 2, 2, 2)),
                                                    (40, 240, 28, 28, 3, 2, (0, 1, 0, 1)), (16, 100, 16, 16, 3, 2, (0, 1, 0, 1)),
                                                    # the whole-image kernel of the 14x14 / 7x7 stride-1 blocks (B = 3: a 7x7 group of four is ragged)
                                                    (80, 480, 14, 14, 3, 1, (1, 1, 1, 1)), (80, 480, 14, 14, 5, 1, (2, 2, 2, 2)),
                                                    (112, 672, 14, 14, 5, 1, (2, 2, 2, 2)), (112, 100, 14, 14, 3, 1, (1, 1, 1, 1)),
                                                    (192, 1152, 7, 7, 5, 1, (2, 2, 2, 2)), (192, 1152, 7, 7, 3, 1, (1, 1, 1, 1)),
                                                    # the block that halves 14 x 14: whole-image kernel, stride 2 (split products only)
                                                    (112, 672, 14, 14, 5, 2, (1, 2, 1, 2)), (112, 100, 14, 14, 5, 2, (1, 2, 1, 2))])
@pytest.mark.parametrize("mode", ["f32", "f32x3"])
def test_fused_expand_dwconv(Cin, Cexp, H, W, ks, s, pad, mode):
    """1x1 expand + BN + swish + depthwise + BN + swish in one kernel == the two torch convolutions; sizes that do not
    divide the kernel's tiles, TF-'same' asymmetric padding, a channel count that is not a multiple of the 32-wide chunk.
    Both with the fp32 matrix-core instruction and with split bf16 products (same fp32 tolerance)."""
    from image_captioning_amd import ops
    d = dev()
    ops.set_compute_dtype(mode)
    try:
        if (H, s) == (14, 2) and mode != "f32x3":     # no fused kernel for this shape on the fp32 instruction: the caller
            assert not ops.expand_dwconv_supported(H, W, Cin, Cexp, ks, s, pad)   # takes the two-kernel path, a direct call raises
            with pytest.raises(ops.IcapError):
                ops.expand_dwconv2d(torch.zeros(1, 14, 14, 112, device=d), torch.zeros(672, 112, device=d), None, None,
                                    torch.zeros(25, 672, device=d), None, None, ks=5, stride=2, pad=(1, 2, 1, 2))
            return
        _fused_expand_dwconv_case(ops, d, Cin, Cexp, H, W, ks, s, pad)
    finally:
        ops.set_compute_dtype("f32")


@pytest.mark.parametrize("Cexp", [96, 80])
def test_fused_expand_dwconv_all_chunk_parameters(Cexp):
    """The 16-channel layer with at most three channel chunks and enough workgroups for one z slice (tiles x images >= 4096)
    runs the kernel form that keeps every chunk's parameters on chip (bias rows in the pad columns of the LDS tile): a batch
    of 700 images of 30 x 30 (ragged tiles on two borders), a full last chunk and a half-empty one."""
    from image_captioning_amd import ops
    ops.set_compute_dtype("f32x3")
    try:
        _fused_expand_dwconv_case(ops, dev(), 16, Cexp, 30, 30, 3, 2, (0, 1, 0, 1), B=700)
    finally:
        ops.set_compute_dtype("f32")


def _fused_expand_dwconv_case(ops, d, Cin, Cexp, H, W, ks, s, pad, B=3):
    x = rnd(B, Cin, H, W, seed=1)
    we = rnd(Cexp, Cin, 1, 1, seed=2, scale=Cin ** -0.5)
    es, eb = torch.rand(Cexp, generator=torch.Generator().manual_seed(3)) + 0.5, rnd(Cexp, seed=4, scale=0.3)
    wd = rnd(Cexp, 1, ks, ks, seed=5, scale=1.0 / ks)
    dsc, dbi = torch.rand(Cexp, generator=torch.Generator().manual_seed(6)) + 0.5, rnd(Cexp, seed=7, scale=0.1)
    e = F.conv2d(x, we) * es[None, :, None, None] + eb[None, :, None, None]
    e = e * torch.sigmoid(e)
    ref = F.conv2d(F.pad(e, (pad[2], pad[3], pad[0], pad[1])), wd, stride=s, groups=Cexp)
    ref = ref * dsc[None, :, None, None] + dbi[None, :, None, None]
    ref = ref * torch.sigmoid(ref)
    assert ops.expand_dwconv_supported(H, W, Cin, Cexp, ks, s, pad)
    y, part = ops.expand_dwconv2d(x.permute(0, 2, 3, 1).contiguous().to(d), we.view(Cexp, Cin).contiguous().to(d), es.to(d), eb.to(d),
                                  wd.view(Cexp, ks * ks).t().contiguous().to(d), dsc.to(d), dbi.to(d), ks=ks, stride=s, pad=pad,
                                  want_partial=True)
    close(y.permute(0, 3, 1, 2), ref, what="expand+dwconv")
    close(part.sum(1) / (ref.shape[2] * ref.shape[3]), ref.mean((2, 3)), what="expand+dwconv partial pool")
    assert not ops.expand_dwconv_supported(14, 14, 112, 672, 3, 2, (0, 1, 0, 1))      # no kernel for this shape in any mode


@pytest.mark.parametrize("H,W", [(56, 56), (45, 61)])
def test_fused_stem_dwconv(H, W):
    """EfficientNet stem (3x3 stride 2, TF-'same') + BN + swish + block 0's depthwise 3x3 + BN + swish in one kernel."""
    from image_captioning_amd import ops
    d = dev()
    B, C = 3, 32
    x = rnd(B, 3, H, W, seed=1)
    ws = rnd(C, 3, 3, 3, seed=2, scale=27 ** -0.5)
    ss, sb = torch.rand(C, generator=torch.Generator().manual_seed(3)) + 0.5, rnd(C, seed=4, scale=0.3)
    wd = rnd(C, 1, 3, 3, seed=5, scale=1.0 / 3)
    dsc, dbi = torch.rand(C, generator=torch.Generator().manual_seed(6)) + 0.5, rnd(C, seed=7, scale=0.1)
    sp = (0, 2 - (H % 2), 0, 2 - (W % 2))                      # TF-'same' for a 3x3 stride-2 convolution
    e = F.conv2d(F.pad(x, (sp[2], sp[3], sp[0], sp[1])), ws, stride=2) * ss[None, :, None, None] + sb[None, :, None, None]
    e = e * torch.sigmoid(e)
    ref = F.conv2d(e, wd, padding=1, groups=C) * dsc[None, :, None, None] + dbi[None, :, None, None]
    ref = ref * torch.sigmoid(ref)
    x4 = ops.nchw_to_nhwc_pad(x.to(d), 4)
    ws4 = F.pad(ws.permute(0, 2, 3, 1), (0, 1)).contiguous()
    y, part = ops.stem_dwconv2d(x4, ws4.to(d), ss.to(d), sb.to(d), wd.view(C, 9).t().contiguous().to(d), dsc.to(d), dbi.to(d),
                                stem_pad=sp, pad=(1, 1, 1, 1), want_partial=True)
    close(y.permute(0, 3, 1, 2), ref, what="stem+dwconv")
    close(part.sum(1) / (ref.shape[2] * ref.shape[3]), ref.mean((2, 3)), what="stem+dwconv partial pool")
    # the planar NCHW image read directly (icap_stem_dwconv_nchw_f32): the same arithmetic on the same values, bit for bit
    y2, part2 = ops.stem_dwconv2d(x.to(d), ws4.to(d), ss.to(d), sb.to(d), wd.view(C, 9).t().contiguous().to(d), dsc.to(d), dbi.to(d),
                                  stem_pad=sp, pad=(1, 1, 1, 1), want_partial=True, nchw=True)
    assert torch.equal(y2, y) and torch.equal(part2, part)
    # the vector-ALU kernel on the planar image and the k-major filter (icap_stem_dwconv_valu_f32): other order of the fp32 sums
    y3, part3 = ops.stem_dwconv2d(x.to(d), ws4.to(d), ss.to(d), sb.to(d), wd.view(C, 9).t().contiguous().to(d), dsc.to(d), dbi.to(d),
                                  stem_pad=sp, pad=(1, 1, 1, 1), want_partial=True, nchw=True, w_stem_k=ops.stem_weight_k(ws4.to(d)))
    assert not torch.equal(y3, y)                                # it is the other kernel
    close(y3.permute(0, 3, 1, 2), ref, what="stem+dwconv (vector ALU)")
    close(part3.sum(1) / (ref.shape[2] * ref.shape[3]), ref.mean((2, 3)), what="stem+dwconv (vector ALU) partial pool")


def test_pools():
    from image_captioning_amd import ops
    d = dev()
    x = rnd(2, 64, 23, 23, seed=1)
    xh = x.permute(0, 2, 3, 1).contiguous().to(d)
    close(ops.maxpool2d(xh, 3, 2, 1).permute(0, 3, 1, 2), F.max_pool2d(x, 3, 2, 1), what="maxpool 3/2/1")
    close(ops.maxpool2d(xh, 2, 2, 0).permute(0, 3, 1, 2), F.max_pool2d(x, 2, 2), what="maxpool 2/2")
    close(ops.adaptive_avgpool2d(xh, 1, 1).permute(0, 3, 1, 2), F.adaptive_avg_pool2d(x, 1), what="global avgpool")
    close(ops.adaptive_avgpool2d(xh, 7, 7).permute(0, 3, 1, 2), F.adaptive_avg_pool2d(x, 7), what="adaptive 7x7")


def test_embedding_layernorm_argmax():
    from image_captioning_amd import ops
    d = dev()
    table, idx, add = rnd(97, 64, seed=1), torch.tensor([0, 5, 96, 5, 33]), rnd(5, 64, seed=2)
    out, err = ops.embedding(table.to(d), idx.to(d), add.to(d))
    assert torch.equal(out.cpu(), table[idx] + add) and int(err.item()) == 0
    _, err = ops.embedding(table.to(d), torch.tensor([97]).to(d))
    assert int(err.item()) == 1
    x, res, g, b = rnd(11, 512, seed=3), rnd(11, 512, seed=4), rnd(512, seed=5), rnd(512, seed=6)
    close(ops.layernorm(x.to(d), g.to(d), b.to(d), res=res.to(d)), F.layer_norm(x + res, (512,), g, b), what="layernorm")
    close(ops.layernorm(x.to(d), g.to(d), b.to(d), res=x.to(d)), F.layer_norm(2 * x, (512,), g, b), what="layernorm 2x")
    logits = rnd(9, 6756, seed=7)
    logits[2, 100] = logits[2, 4000] = 50.0        # tie: the first index wins
    logits[3, 6755] = 60.0
    idx, gap = ops.argmax_rows(logits.to(d), want_gap=True)
    assert idx.cpu().tolist() == logits.argmax(1).tolist() and idx[2].item() == 100
    t2 = torch.topk(logits, 2, dim=1).values
    close(gap, t2[:, 0] - t2[:, 1], what="top-2 gap")


@pytest.mark.parametrize("Lq,Lk,N,Nk,E,heads", [(5, 5, 1, 1, 512, 8), (30, 3, 1, 1, 64, 4), (4, 6, 3, 3, 512, 2), (7, 12, 1, 1, 512, 4)])
def test_mha_core(Lq, Lk, N, Nk, E, heads):
    from image_captioning_amd import ops
    d = dev()
    q, k, v = rnd(Lq, N, E, seed=1), rnd(Lk, Nk, E, seed=2), rnd(Lk, Nk, E, seed=3)
    hd = E // heads
    kk, vv = k.expand(Lk, N, E), v.expand(Lk, N, E)
    Q = q.reshape(Lq, N * heads, hd).transpose(0, 1) / hd ** 0.5
    K = kk.reshape(Lk, N * heads, hd).transpose(0, 1)
    V = vv.reshape(Lk, N * heads, hd).transpose(0, 1)
    ref = (torch.softmax(Q @ K.transpose(1, 2), -1) @ V).transpose(0, 1).reshape(Lq, N, E)
    close(ops.mha_core(q.to(d), k.to(d), v.to(d), heads), ref, what="mha core")


@pytest.mark.parametrize("B", [1, 5, 64])
def test_bahdanau_attention(B):
    from image_captioning_amd import ops
    d = dev()
    A, E = 256, 512
    u, w, aw, feat = rnd(B, A, seed=1), rnd(B, A, seed=2), rnd(A, seed=3, scale=A ** -0.5), rnd(B, E, seed=4)
    scores = (torch.tanh(u[None, :, :] + w[:, None, :]) * aw).sum(-1) + 0.3
    alpha = torch.softmax(scores, 1)
    a, c = ops.bahdanau_attention(u.to(d), w.to(d), aw.to(d), 0.3, feat.to(d))
    close(a, alpha, what="alpha")
    close(c, alpha @ feat, what="context")


def test_pointwise_cells():
    from image_captioning_amd import ops
    from oracle import torch_ref as R
    d = dev()
    B, H, I = 5, 64, 32
    w_ih, w_hh, b_ih, b_hh = rnd(4 * H, I, seed=1, scale=0.2), rnd(4 * H, H, seed=2, scale=0.2), rnd(4 * H, seed=3), rnd(4 * H, seed=4)
    x, h, c = rnd(B, I, seed=5), rnd(B, H, seed=6), rnd(B, H, seed=7)
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    h2, c2 = ops.lstm_pointwise(gates.to(d), c.to(d))
    rh, rc = R.lstm_cell(w_ih, w_hh, b_ih, b_hh, x, h, c)
    close(h2, rh, what="lstm h"); close(c2, rc, what="lstm c")
    g_ih, g_hh = w_ih[:3 * H], w_hh[:3 * H]
    gi, gh = x @ g_ih.t() + b_ih[:3 * H], h @ g_hh.t() + b_hh[:3 * H]
    close(ops.gru_pointwise(gi.to(d), gh.to(d), h.to(d)), R.gru_cell(g_ih, g_hh, b_ih[:3 * H], b_hh[:3 * H], x, h), what="gru h")


@pytest.mark.parametrize("NB", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("N,K", [(6756, 512), (512, 2048), (81, 512), (2048, 512)])
def test_gemv(NB, N, K):
    from image_captioning_amd import ops
    d = dev()
    W, b, x = rnd(N, K, seed=1, scale=K ** -0.5), rnd(N, seed=2), rnd(NB, K, seed=3)
    ref = x @ W.t() + b
    y, pv, pi = ops.gemv(W.to(d), x.to(d), b.to(d), want_partials=True)
    close(y, ref, what="gemv")
    best = pv.cpu().max(1)
    got_idx = pi.cpu().gather(1, best.indices[:, None])[:, 0]
    assert got_idx.tolist() == ref.argmax(1).tolist()
    close(ops.gemv(W.to(d), x.to(d), b.to(d), act=ops.ACT_RELU), torch.relu(ref), what="gemv relu")


@pytest.mark.parametrize("NB", [1, 4, 8])
@pytest.mark.parametrize("I,H", [(512, 512), (1024, 512), (32, 64)])
def test_fused_cells(NB, I, H):
    from image_captioning_amd import ops
    from oracle import torch_ref as R
    d = dev()
    w_ih, w_hh = rnd(4 * H, I, seed=1, scale=I ** -0.5), rnd(4 * H, H, seed=2, scale=H ** -0.5)
    b_ih, b_hh = rnd(4 * H, seed=3, scale=0.1), rnd(4 * H, seed=4, scale=0.1)
    x, h, c = rnd(NB, I, seed=5), rnd(NB, H, seed=6, scale=0.5), rnd(NB, H, seed=7, scale=0.5)
    h2, c2 = ops.lstm_cell(w_ih.to(d), w_hh.to(d), b_ih.to(d), b_hh.to(d), x.to(d), h.to(d), c.to(d))
    rh, rc = R.lstm_cell(w_ih, w_hh, b_ih, b_hh, x, h, c)
    close(h2, rh, what="lstm_cell h"); close(c2, rc, what="lstm_cell c")
    g = slice(0, 3 * H)
    h3 = ops.gru_cell(w_ih[g].contiguous().to(d), w_hh[g].contiguous().to(d), b_ih[g].contiguous().to(d),
                      b_hh[g].contiguous().to(d), x.to(d), h.to(d))
    close(h3, R.gru_cell(w_ih[g], w_hh[g], b_ih[g], b_hh[g], x, h), what="gru_cell h")


@pytest.mark.parametrize("M,K,N", [(256, 512, 6756), (5, 64, 100), (70, 512, 84), (1024, 512, 6756), (2048, 512, 6756), (1024, 48, 6756)])
def test_linear_argmax_and_next_token(M, K, N):
    from image_captioning_amd import ops
    d = dev()
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = x @ w.t() + b
    pv, pi, y = ops.linear_argmax(x.to(d), w.to(d), b.to(d), want_logits=True)
    close(y, ref, what="linear_argmax logits")
    E = 64
    table, add, g, be = rnd(N, E, seed=4), rnd(M, E, seed=5), rnd(E, seed=6), rnd(E, seed=7)
    ids = torch.full((M, 3), -1, dtype=torch.int64, device=d)
    out = ops.next_token(pv, pi, ids, 1, table.to(d), add.to(d), g.to(d), be.to(d))
    want = ref.argmax(1)
    assert ids[:, 1].cpu().tolist() == want.tolist() and (ids[:, 0] == -1).all() and (ids[:, 2] == -1).all()
    close(out, F.layer_norm(table[want] + add, (E,), g, be), what="next_token rows")
    pv2, pi2, none = ops.linear_argmax(x.to(d), w.to(d), b.to(d))
    assert none is None
    plain = ops.next_token(pv2, pi2, ids, 2, table.to(d))
    assert ids[:, 2].cpu().tolist() == want.tolist() and torch.equal(plain.cpu(), table[want])


def test_argument_errors_are_raised():
    from image_captioning_amd import ops
    d = dev()
    with pytest.raises(ops.IcapError):
        ops.linear(torch.zeros(4, 8), torch.zeros(3, 8))                       # CPU tensors: no fallback
    with pytest.raises(ops.IcapError):
        ops.conv2d(torch.zeros(1, 4, 4, 6, device=d), torch.zeros(8, 1, 1, 6, device=d))   # Cin % 4
    with pytest.raises(ops.IcapError):
        ops.gemv(torch.zeros(16, 8, device=d), torch.zeros(9, 8, device=d))    # more than 8 lock-step rows


@pytest.mark.parametrize("M,K,N,act,use_res,use_gate,cfg", [
    (300, 512, 6756, 0, False, False, -1), (1000, 112, 672, 2, False, False, 0), (392, 672, 112, 0, True, True, 1),
    (50, 16, 96, 2, False, False, 2), (98, 1152, 320, 0, False, True, -1), (7, 2048, 512, 1, True, False, -1),
    (260, 40, 240, 2, False, False, 0)])
def test_linear_bf16(M, K, N, act, use_res, use_gate, cfg):
    """bf16 products / fp32 accumulation: exact against the same computation on bf16-rounded operands (the rounding is
    the only difference from the fp32 kernel), and within the stated bf16 tolerance of the unrounded result."""
    from image_captioning_amd import ops
    d = dev()
    hw = 49 if use_gate else 1
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    scale = torch.rand(N, generator=torch.Generator().manual_seed(4)) + 0.5
    res = rnd(M, N, seed=5) if use_res else None
    gate = torch.rand(M // hw, K, generator=torch.Generator().manual_seed(6)) if use_gate else None
    xg = x * gate.repeat_interleave(hw, 0) if use_gate else x
    w16 = ops.to_bf16(w.to(d))
    assert torch.equal(w16.cpu(), w.to(torch.bfloat16))                       # round to nearest even, like torch
    got = ops.linear_bf16(x.to(d), w16, b.to(d), scale.to(d), act=act, residual=res.to(d) if use_res else None,
                          gate=gate.to(d) if use_gate else None, hw=hw, cfg=cfg)

    def ref(xa, wa):
        y = (xa.double() @ wa.double().t()) * scale.double() + b.double()
        if use_res:
            y = y + res.double()
        return act_ref(y, act).float()
    exact = ref(xg.to(torch.bfloat16).float(), w.to(torch.bfloat16).float())
    close(got, exact, rtol=1e-4, atol=1e-4, what="linear_bf16 vs bf16-rounded operands")
    close(got, ref(xg, w), rtol=2e-2, atol=2e-2, what="linear_bf16 vs fp32 (bf16 tolerance: 2e-2)")


@pytest.mark.parametrize("M,K,N,act,use_res,use_gate,cfg", [
    (1024, 512, 6756, 0, False, False, -1), (300, 512, 6756, 0, False, False, 2), (1000, 112, 672, 2, False, False, 0),
    (392, 672, 112, 0, True, True, 1), (98, 1152, 320, 0, False, True, -1), (7, 2048, 512, 1, True, False, -1),
    (260, 40, 240, 2, False, False, 0), (200, 2048, 512, 0, True, False, 1),
    (300, 1000, 200, 1, True, False, 2), (130, 32, 132, 0, False, False, 0), (129, 24, 4, 3, False, False, 1)])
def test_linear_f32x3_is_fp32_accurate(M, K, N, act, use_res, use_gate, cfg):
    """Six bf16 products per multiply from exact three-way splits: the result must be as close to the float64 product as
    an fp32 computation is (the tolerance is a small multiple of what the fp32 MFMA kernel itself leaves)."""
    from image_captioning_amd import ops
    d = dev()
    hw = 49 if use_gate else 1
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    scale = torch.rand(N, generator=torch.Generator().manual_seed(4)) + 0.5
    res = rnd(M, N, seed=5) if use_res else None
    gate = torch.rand(M // hw, K, generator=torch.Generator().manual_seed(6)) if use_gate else None
    xg = x * gate.repeat_interleave(hw, 0) if use_gate else x
    w3 = ops.split3(w.to(d))
    planes = w3.float().cpu()
    assert torch.equal(planes[0], w.to(torch.bfloat16).float())
    assert torch.equal(planes.double().sum(0).float(), w) or (planes.double().sum(0) - w.double()).abs().max() <= 2.0 ** -24 * w.abs().max()
    got = ops.linear_f32x3(x.to(d), w3, b.to(d), scale.to(d), act=act, residual=res.to(d) if use_res else None,
                           gate=gate.to(d) if use_gate else None, hw=hw, cfg=cfg).cpu().double()
    ref = (xg.double() @ w.double().t()) * scale.double() + b.double()
    if use_res:
        ref = ref + res.double()
    ref = act_ref(ref, act)
    f32 = ops.linear(xg.to(d), w.to(d), b.to(d), scale.to(d), act=act, residual=res.to(d) if use_res else None).cpu().double()
    err, err32 = (got - ref).abs().max().item(), (f32 - ref).abs().max().item()
    assert err <= max(4 * err32, 2e-6 * ref.abs().max().item()), (err, err32)


@pytest.mark.parametrize("M,K,N", [(1024, 512, 6756), (70, 512, 84), (256, 2048, 6756)])
def test_linear_f32x3_argmax(M, K, N):
    from image_captioning_amd import ops
    d = dev()
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = x.double() @ w.double().t() + b.double()
    pv, pi, y = ops.linear_f32x3_argmax(x.to(d), ops.split3(w.to(d)), b.to(d), want_logits=True)
    assert (y.cpu().double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
    ids = torch.full((M, 1), -1, dtype=torch.int64, device=d)
    ops.next_token(pv, pi, ids, 0)
    top2 = torch.topk(ref, 2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(ids[:, 0].cpu()[clear], ref.argmax(1)[clear])
    pv2, pi2, none = ops.linear_f32x3_argmax(x.to(d), ops.split3(w.to(d)), b.to(d))
    assert none is None and torch.equal(pv2, pv) and torch.equal(pi2, pi)


def test_linear_f32x3_with_presplit_activations():
    """The split-product GEMM fed with split3(x) planes gives bit-identical results to splitting x inside the kernel
    (the same three planes either way), for the plain and the arg-max form."""
    from image_captioning_amd import ops
    d = dev()
    M, K, Nn = 300, 512, 1000
    x, w, b = rnd(M, K, seed=1).to(d), rnd(Nn, K, seed=2, scale=K ** -0.5).to(d), rnd(Nn, seed=3).to(d)
    w3, x3 = ops.split3(w), ops.split3(x)
    for cfg in (0, 1, 2):
        assert torch.equal(ops.linear_f32x3(x3, w3, b, act=ops.ACT_RELU, cfg=cfg), ops.linear_f32x3(x, w3, b, act=ops.ACT_RELU, cfg=cfg))
    pv0, pi0, lg0 = ops.linear_f32x3_argmax(x, w3, b, want_logits=True)
    pv1, pi1, lg1 = ops.linear_f32x3_argmax(x3, w3, b, want_logits=True)
    assert torch.equal(pv0, pv1) and torch.equal(pi0, pi1) and torch.equal(lg0, lg1)
    with pytest.raises(ops.IcapError):
        ops.linear_f32x3(x3, w3, gate=torch.ones(M, K, device=d))


@pytest.mark.parametrize("M,K,N,cfg", [(130, 64, 132, 0), (130, 64, 132, 2), (70, 512, 84, 1), (130, 40, 132, 0)])
def test_linear_f32x3_partial_tiles_take_nothing_from_row_zero(M, K, N, cfg):
    """Tiles that overhang M or N stage row 0 in place of the missing rows when every K slice is whole (K % 32 == 0; zeros
    otherwise).  Whatever row 0 holds - here NaN - stays in row 0 of x / column 0 of the result: every other output is bit
    for bit what it is with a finite row 0, for the plain form, pre-split activations and the arg-max partials."""
    from image_captioning_amd import ops
    d = dev()
    x, w, b = rnd(M, K, seed=1).to(d), rnd(N, K, seed=2, scale=K ** -0.5).to(d), rnd(N, seed=3).to(d)
    xn, wn = x.clone(), w.clone()
    xn[0] = float("nan")
    wn[0] = float("nan")
    w3, wn3 = ops.split3(w), ops.split3(wn)
    clean = ops.linear_f32x3(x, w3, b, cfg=cfg)
    assert torch.isfinite(clean).all()
    for xs, ws in ((xn, w3), (x, wn3), (xn, wn3), (ops.split3(xn), wn3)):
        got = ops.linear_f32x3(xs, ws, b, cfg=cfg)
        assert torch.equal(got[1:, 1:], clean[1:, 1:])
    if N % 4 == 0:
        pv0, pi0, _ = ops.linear_f32x3_argmax(x, w3, b)
        for xs in (xn, ops.split3(xn)):
            pv1, pi1, _ = ops.linear_f32x3_argmax(xs, w3, b)
            assert torch.equal(pv1[1:], pv0[1:]) and torch.equal(pi1[1:], pi0[1:])


@pytest.mark.parametrize("rows,E", [(11, 512), (1024, 512), (5, 96), (3, 1000)])
def test_layernorm_writes_its_result_split(rows, E):
    """ops.layernorm(split=True): the planes are bit for bit split3 of the fp32 LayerNorm result, and they add back up to it."""
    from image_captioning_amd import ops
    d = dev()
    x, res, g, b = rnd(rows, E, seed=3).to(d), rnd(rows, E, seed=4).to(d), rnd(E, seed=5).to(d), rnd(E, seed=6).to(d)
    y = ops.layernorm(x, g, b, res=res)
    y3 = ops.layernorm(x, g, b, res=res, split=True)
    assert y3.dtype == torch.bfloat16 and tuple(y3.shape) == (3, rows, E)
    assert torch.equal(y3.view(torch.int16), ops.split3(y).view(torch.int16))
    back = y3[0].double() + y3[1].double() + y3[2].double()
    assert float((back - y.double()).abs().max()) <= 2e-7 * float(y.abs().max())


@pytest.mark.parametrize("M,K,N,hw,use_res,use_gate", [
    (3136 * 3, 32, 16, 3136, False, True), (784 * 5, 96, 24, 784, False, True), (784 * 4 + 0, 144, 24, 784, True, True),
    (49 * 64 + 17, 144, 28, 49, False, True), (49 * 40, 96, 32, 49, True, True), (1000, 144, 32, 1, True, False),
    (37, 32, 4, 37, False, True), (65, 144, 32, 1, False, False), (33 * 50, 96, 20, 33, True, True), (31, 32, 16, 1, True, False)])
def test_linear_thin(M, K, N, hw, use_res, use_gate):
    """The streaming kernel of the thin projection layers against a float64 product: every (K, width) instantiation, gate
    rows that change inside a 32-row block (hw not a multiple of 32), ragged last block, residual."""
    from image_captioning_amd import ops
    d = dev()
    if use_gate and M % hw:
        M = M // hw * hw
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    scale = torch.rand(N, generator=torch.Generator().manual_seed(4)) + 0.5
    res = rnd(M, N, seed=5) if use_res else None
    gate = torch.rand(M // hw, K, generator=torch.Generator().manual_seed(6)) if use_gate else None
    xg = x * gate.repeat_interleave(hw, 0) if use_gate else x
    assert ops.linear_thin_supported(K, N)
    got = ops.linear_thin(x.to(d), w.to(d), b.to(d), scale.to(d), residual=res.to(d) if use_res else None,
                          gate=gate.to(d) if use_gate else None, hw=hw).cpu().double()
    ref = (xg.double() @ w.double().t()) * scale.double() + b.double()
    if use_res:
        ref = ref + res.double()
    f32 = ops.linear(xg.to(d), w.to(d), b.to(d), scale.to(d), residual=res.to(d) if use_res else None).cpu().double()
    err, err32 = (got - ref).abs().max().item(), (f32 - ref).abs().max().item()
    assert err <= max(4 * err32, 2e-6 * ref.abs().max().item()), (err, err32)
    assert not ops.linear_thin_supported(64, 16) and not ops.linear_thin_supported(32, 40) and not ops.linear_thin_supported(240, 40)
    with pytest.raises(ops.IcapError):
        ops.linear_thin(torch.zeros(64, 64, device=d), torch.zeros(16, 64, device=d))


@pytest.mark.parametrize("M,K,N,hw,use_res,use_gate", [
    (196 * 7, 480, 80, 196, True, True), (196 * 5 + 0, 672, 112, 196, True, True), (196 * 3, 240, 80, 196, False, True),
    (49 * 30, 1152, 192, 49, True, True), (49 * 11, 1152, 320, 49, False, True), (49 * 9, 672, 192, 49, False, True),
    (1000, 96, 24, 1, True, False), (129, 48, 4, 1, False, False), (127, 192, 36, 127, False, True), (43 * 12, 144, 200, 43, True, True)])
def test_linear_f32x3_tall(M, K, N, hw, use_res, use_gate):
    """cfg 3 of the split-product layer (activations kept in registers over the whole contraction) against a float64 product
    and against the tiled split-product kernel: 48- and 96-wide chunks, one to six column blocks and two column tiles, images
    that change inside a tile (hw = 49, 43), ragged last tile, residual."""
    from image_captioning_amd import ops
    d = dev()
    x, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    scale = torch.rand(N, generator=torch.Generator().manual_seed(4)) + 0.5
    res = rnd(M, N, seed=5) if use_res else None
    gate = torch.rand(M // hw, K, generator=torch.Generator().manual_seed(6)) if use_gate else None
    xg = x * gate.repeat_interleave(hw, 0) if use_gate else x
    assert ops.linear_f32x3_tall_supported(K, N, hw, use_gate)
    w3 = ops.split3(w.to(d))
    kw = dict(residual=res.to(d) if use_res else None, gate=gate.to(d) if use_gate else None, hw=hw)
    got = ops.linear_f32x3(x.to(d), w3, b.to(d), scale.to(d), cfg=3, **kw).cpu().double()
    tiled = ops.linear_f32x3(x.to(d), w3, b.to(d), scale.to(d), cfg=2, **kw).cpu().double()
    ref = (xg.double() @ w.double().t()) * scale.double() + b.double()
    if use_res:
        ref = ref + res.double()
    err, err_t = (got - ref).abs().max().item(), (tiled - ref).abs().max().item()
    assert err <= max(4 * err_t, 2e-6 * ref.abs().max().item()), (err, err_t)
    assert not ops.linear_f32x3_tall_supported(100, 16) and not ops.linear_f32x3_tall_supported(96, 18)
    assert not ops.linear_f32x3_tall_supported(96, 16, 36, True)
    with pytest.raises(ops.IcapError):
        ops.linear_f32x3(torch.zeros(64, 64, device=d), ops.split3(torch.zeros(16, 64, device=d)), cfg=3)


@pytest.mark.parametrize("M,K,N,ks,cfg,use_res,use_gate,act", [
    (1024, 2048, 512, 4, 2, True, False, 0), (300, 1024, 260, 2, 1, False, False, 1), (130, 512, 68, 8, 2, True, False, 2),
    (49 * 6, 1152, 192, 4, 0, True, True, 0), (64, 1024, 64, 16, 2, False, False, 0)])
def test_linear_f32x3_splitk(M, K, N, ks, cfg, use_res, use_gate, act):
    """Split-K form of the split-product layer (slices of the contraction in grid.z, partial products through a workspace, ordered
    reduction with the epilogue) against the one-pass kernel: ragged tiles, every tile configuration, gate, residual, activation."""
    from image_captioning_amd import ops
    d = dev()
    hw = 49 if use_gate else 1
    x, w, b = rnd(M, K, seed=1).to(d), rnd(N, K, seed=2, scale=K ** -0.5).to(d), rnd(N, seed=3).to(d)
    scale = (torch.rand(N, generator=torch.Generator().manual_seed(4)) + 0.5).to(d)
    res = rnd(M, N, seed=5).to(d) if use_res else None
    gate = torch.rand(M // hw, K, generator=torch.Generator().manual_seed(6)).to(d) if use_gate else None
    w3 = ops.split3(w)
    one = ops.linear_f32x3(x, w3, b, scale, act=act, residual=res, gate=gate, hw=hw, cfg=2)
    got = ops.linear_f32x3(x, w3, b, scale, act=act, residual=res, gate=gate, hw=hw, cfg=(ks << 4) | cfg)
    again = ops.linear_f32x3(x, w3, b, scale, act=act, residual=res, gate=gate, hw=hw, cfg=(ks << 4) | cfg)
    assert torch.equal(got, again)                              # fixed reduction order
    tol = 4e-6 * float(one.abs().max())                         # the same products, summed in another order
    assert float((got - one).abs().max()) <= tol
    with pytest.raises(ops.IcapError):                          # K must split into whole 32-wide steps
        ops.linear_f32x3(x[:, :K - 32].contiguous(), ops.split3(w[:, :K - 32].contiguous()), cfg=(16 << 4) | 2)
