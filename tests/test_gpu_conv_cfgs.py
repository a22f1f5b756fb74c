"""GPU: every tile configuration of the implicit-GEMM kernel (the set the per-shape tuner picks from, so any of them can
end up in a timed or production run) against the fp32 torch CPU convolution, plus the tuner's own choice."""
import functools

import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_kernels import act_ref, close, dev, rnd

pytestmark = pytest.mark.gpu

# (KSPLIT, KU, PF, FRAG) of every configuration index: decides which shapes a configuration must reject.  A K step
# stages 16 * KSPLIT * KU elements; PF > 1 needs the step count to be a multiple of PF, FRAG an even count >= 4.
CFG_TRAITS = [
    (1, 1, 1, 0), (1, 1, 1, 0), (1, 1, 1, 0), (1, 1, 1, 0), (1, 1, 1, 0), (4, 1, 1, 0), (1, 1, 1, 0), (1, 1, 1, 0),
    (1, 1, 1, 0), (1, 1, 1, 0), (1, 1, 1, 0), (1, 2, 1, 0), (4, 2, 1, 0), (1, 2, 1, 0), (1, 2, 1, 0), (2, 2, 1, 0),
    (2, 1, 1, 0), (2, 2, 1, 0), (4, 1, 1, 0), (2, 2, 1, 0), (4, 2, 1, 0), (8, 1, 1, 0),
    (1, 2, 2, 0), (4, 1, 2, 0), (1, 1, 2, 0), (1, 1, 4, 0), (1, 1, 8, 0), (1, 1, 4, 0), (4, 1, 4, 0),
    (1, 1, 1, 1), (1, 1, 1, 1), (1, 1, 1, 1), (1, 2, 1, 1), (4, 1, 1, 1), (1, 1, 1, 1), (1, 1, 1, 1),
]

SHAPES = [
    # B, H, W, Cin, Cout, KS, stride, pad(t,b,l,r), act, residual, gate
    (3, 10, 10, 512, 200, 1, 1, (0, 0, 0, 0), 0, True, True),       # K=512: 32 steps (every pipelined form runs); M=300, Cout=200 ragged
    (2, 13, 13, 80, 72, 1, 1, (0, 0, 0, 0), 2, False, False),       # K=80 < 128: 5 steps of 16 (odd), split-K forms mask most of a step
    (2, 12, 11, 24, 136, 3, 1, (1, 1, 1, 1), 1, True, False),       # 3x3 gather, K=216: 14 steps (even, not a multiple of 4)
    (2, 23, 23, 4, 64, 7, 2, (3, 3, 3, 3), 1, False, False),        # 7x7 stride-2 stem gather, K=196: 13 steps (odd)
    (3, 17, 15, 16, 40, 3, 2, (0, 1, 0, 1), 3, False, False),       # TF-same asymmetric 3x3 stride 2, K=144: 9 steps
    (5, 20, 20, 256, 96, 1, 1, (0, 0, 0, 0), 4, False, True),       # K=256: 16 steps; M=2000; SE gate; tanh
    (1024, 1, 1, 512, 6756, 1, 1, (0, 0, 0, 0), 0, False, False),   # the benchmark's vocabulary projection (weights outweigh rows)
    (3, 21, 19, 32, 16, 1, 1, (0, 0, 0, 0), 0, False, True),        # a thin project convolution: K=32 (2 steps / 1 step of 32), Cout=16, SE gate
    (2, 15, 15, 96, 24, 1, 1, (0, 0, 0, 0), 0, True, True),         # K=96: 6 steps / 3 steps of 32; residual + gate
    (7, 1, 1, 2048, 512, 1, 1, (0, 0, 0, 0), 1, True, False),       # few-row decoder GEMM: tiles mostly padding
]


def rejects(cfg, K):
    ks, ku, pf, frag = CFG_TRAITS[cfg]
    step = 16 * ks * ku
    nk = (K + step - 1) // step
    return (pf > 1 and nk % pf != 0) or (frag and (nk < 4 or nk % 2))


@functools.lru_cache(maxsize=None)
def problem(i):
    B, H, W, Cin, Cout, K, s, pad, act, use_res, use_gate = SHAPES[i]
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
    ref = act_ref(ref, act).permute(0, 2, 3, 1).contiguous()
    args = dict(x=x.permute(0, 2, 3, 1).contiguous().to(d), w=w.permute(0, 2, 3, 1).contiguous().to(d), scale=scale.to(d), bias=bias.to(d),
                residual=res.permute(0, 2, 3, 1).contiguous().to(d) if use_res else None, gate=gate.to(d) if use_gate else None)
    return args, ref


def test_trait_table_covers_every_configuration():
    from image_captioning_amd import ops
    assert ops.conv2d_num_configs() == len(CFG_TRAITS)


@pytest.mark.parametrize("shape", range(len(SHAPES)))
def test_every_tile_configuration(shape):
    """All configurations on one problem: each either reproduces the CPU convolution or refuses the shape for the
    documented reason (K-step parity of its pipelined loop) - never anything else.  The output buffer starts as NaN, so a
    tile that is not written fails the comparison."""
    from image_captioning_amd import ops
    B, H, W, Cin, Cout, K, s, pad, act, _, _ = SHAPES[shape]
    a, ref = problem(shape)
    ran = 0
    for cfg in range(ops.conv2d_num_configs()):
        out = torch.full(ref.shape, float("nan"), device=dev())
        if rejects(cfg, K * K * Cin):
            with pytest.raises(ops.IcapError):
                ops.conv2d(a["x"], a["w"], a["scale"], a["bias"], residual=a["residual"], gate=a["gate"], stride=s, pad=pad, act=act, out=out, cfg=cfg)
            continue
        y = ops.conv2d(a["x"], a["w"], a["scale"], a["bias"], residual=a["residual"], gate=a["gate"], stride=s, pad=pad, act=act, out=out, cfg=cfg)
        close(y, ref, what=f"cfg {cfg} on {SHAPES[shape]}")
        ran += 1
    assert ran >= 22        # the 22 plain configurations run every shape


def test_configuration_argument_errors():
    from image_captioning_amd import ops
    a, _ = problem(1)
    B, H, W, Cin, Cout, K, s, pad, act, _, _ = SHAPES[1]
    for bad in (-1, ops.conv2d_num_configs()):
        with pytest.raises(ops.IcapError):
            ops.conv2d(a["x"], a["w"], a["scale"], a["bias"], stride=s, pad=pad, act=act, cfg=bad)
    d = dev()
    with pytest.raises(ops.IcapError):      # Cout % 4 != 0 never takes a tile configuration
        ops.conv2d(torch.zeros(2, 1, 1, 64, device=d), torch.zeros(81, 1, 1, 64, device=d), cfg=0)


@pytest.mark.parametrize("shape", range(len(SHAPES)))
def test_tuned_choice(shape):
    """The tuning entry point (what a warm-up pass with autotune on calls) and the plain entry point afterwards, which
    then runs the remembered winner: both against the CPU convolution."""
    from image_captioning_amd import _native as N, ops
    B, H, W, Cin, Cout, K, s, pad, act, _, _ = SHAPES[shape]
    a, ref = problem(shape)
    ops.set_autotune(True)
    try:
        ops._tuned_shapes.discard((B, H, W, Cin, Cout, K, K, s, ref.shape[1], ref.shape[2]))
        y0 = ops.conv2d(a["x"], a["w"], a["scale"], a["bias"], residual=a["residual"], gate=a["gate"], stride=s, pad=pad, act=act)
    finally:
        ops.set_autotune(False)
    picked = N.load().icap_conv2d_tuned_config(B, H, W, Cin, Cout, K, K, s, ref.shape[1], ref.shape[2])
    assert 0 <= picked < ops.conv2d_num_configs() and not rejects(picked, K * K * Cin)
    close(y0, ref, what=f"tuning call, picked cfg {picked}")
    y1 = ops.conv2d(a["x"], a["w"], a["scale"], a["bias"], residual=a["residual"], gate=a["gate"], stride=s, pad=pad, act=act,
                    out=torch.full(ref.shape, float("nan"), device=dev()))
    close(y1, ref, what=f"after tuning, cfg {picked}")
