"""Tensor-level wrappers over the C ABI.

PyTorch is used for device memory and the current HIP stream only; every
arithmetic operation below runs in ``libicap_hip.so``.  Inputs must live on a HIP
device (``tensor.is_cuda``), be fp32 and contiguous - anything else raises, there
is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N
from ._native import ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SWISH, ACT_TANH, IcapError  # noqa: F401


# The raw handle of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object
# and walks the device-index helpers on every call (about 9 us, a third of the host time of a small-batch training
# step); the two C entry points below return the same handle in a fraction of a microsecond.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


# When True, conv2d / linear time every tile configuration the first time they see a shape (this synchronises
# the stream once per new shape) and use the measured winner afterwards.  Enable while warming up a fixed workload.
AUTOTUNE = False
_tuned_shapes = set()


COMPUTE_DTYPE = "f32"


def set_compute_dtype(name: str) -> None:
    """"f32" (default): every product in fp32 on the fp32 matrix-core instruction.  "f32x3": the MFMA-bound Linear / 1x1
    convolution layers compute fp32-ACCURATE products on the bf16 matrix cores (exact three-way bf16 splits of both
    operands, six products per multiply, fp32 accumulation: linear_f32x3) - same results to a few fp32 roundings, the
    bf16 instruction's 16x flop rate divided by six; layers it does not speed up stay on the fp32 kernel.  "bf16": the 1x1 convolutions of the EfficientNet encoder and the wide
    decoder GEMMs (fc1, vocabulary projection) of the batched greedy path multiply bf16-rounded operands on the bf16
    matrix cores and accumulate in fp32; every tensor in memory stays fp32."""
    global COMPUTE_DTYPE
    if name not in ("f32", "f32x3", "bf16"):
        raise IcapError(f"set_compute_dtype: unknown dtype {name!r}")
    COMPUTE_DTYPE = name


def compute_dtype() -> str:
    return COMPUTE_DTYPE


# Training-step state, set for the duration of one forward + backward by `train_step` (training.Trainer.step does).
# TRAIN_DTYPE "bf16": every Linear / 1x1-convolution product of the step's forward pass and of its input gradients
# (dX = dY W) multiplies bf16-rounded operands on the bf16 matrix cores with fp32 accumulation; parameters, activations,
# gradients and optimizer state stay fp32 in memory, and the weight gradients (dW = dY^T X) are fp32 products.
# TRAIN_DTYPE "f32x3": the same layers (forward and input gradients) as fp32-ACCURATE split products on the bf16 matrix
# cores (linear_f32x3; the weight's three bf16 planes are made once per step), chosen per shape against the fp32 kernel
# exactly as in the captioning path's "f32x3" mode; results equal the fp32 step's to a few fp32 roundings.
# _step_cache: per-step copies derived from the weights (bf16 copies, transposed copies), built the first time a
# weight is used in the step - the LSTM weights are used once per time step - and dropped when the step ends, before
# the optimizer changes the weights.  An entry keeps its source tensor alive, so an address cannot be re-used by
# another tensor while the entry exists.
TRAIN_DTYPE = "f32"
_step_cache = None
# GEMMs with fewer rows than this stay on the fp32 kernel in a bf16 step: the per-time-step decoder GEMMs (M = batch) have
# too few output tiles to fill the chip and run faster on the fp32 kernel's split-K configurations (measured at M = 256:
# 21.7 us fp32 against 34.8 us bf16 for [256, 2048] x [2048, 512]); the 1x1 convolutions (M = batch x pixels) and the
# vocabulary projection (M = batch x tokens) are where bf16 pays (2-3.5x on the layers with K >= 192).
TRAIN_BF16_MIN_ROWS = 1024


class train_step:
    """Context of one training step's forward + backward: `with ops.train_step("bf16"): ...`."""

    def __init__(self, dtype: str = "f32"):
        if dtype not in ("f32", "f32x3", "bf16"):
            raise IcapError(f"train_step: unknown dtype {dtype!r}")
        self.dtype = dtype

    def __enter__(self):
        global TRAIN_DTYPE, _step_cache
        self.saved = (TRAIN_DTYPE, _step_cache)
        TRAIN_DTYPE, _step_cache = self.dtype, {}
        return self

    def __exit__(self, *exc):
        global TRAIN_DTYPE, _step_cache
        TRAIN_DTYPE, _step_cache = self.saved
        return False


def step_cached(kind: str, src: torch.Tensor, make):
    """make() once per (kind, src) within a training step; outside a step it is simply make()."""
    if _step_cache is None:
        return make()
    key = (kind, src.data_ptr(), tuple(src.shape), src.stride(0))
    hit = _step_cache.get(key)
    if hit is None:
        hit = _step_cache[key] = (src, make())
    return hit[1]


# "f32x3" mode: which kernel runs a [M, K] x [N, K] layer.  Without tuning: the split kernel when the contraction is long
# enough for the matrix cores to be the bound (measured cross-over), the fp32 kernel otherwise.  With AUTOTUNE on, the
# first call of a shape times the fp32 kernel and the split kernel's tile configurations and remembers the fastest.
X3_MIN_K = 160
X3_MIN_TILES = 128
_x3_choice: dict = {}
_X3_FP32 = -2
_X3_USE_TUNED = True         # False inside `with untuned()`: the remembered choices are ignored, as conv2d's are


def linear_f32x3_tall_supported(K: int, Nn: int, hw: int = 1, gated: bool = False) -> bool:
    return bool(N.load().icap_linear_f32x3_tall_supported(K, Nn, hw, int(gated)))


def _x3_pick(M, K, Nn, run_fp32, run_split, tall=False, splitk=()):
    """tall: the layer can also run on the tall-activation kernel (cfg 3); splitk: split-K configurations ((slices << 4) | tile)
    worth trying - both are picked by the tuner only."""
    key = (M, K, Nn, tall, splitk)
    c = _x3_choice.get(key) if _X3_USE_TUNED else None
    if c is not None:
        return c
    if K % 8 or Nn % 4:
        return _X3_FP32
    if not AUTOTUNE:
        tiles = ((M + 127) // 128) * ((Nn + 127) // 128)
        return -1 if (K >= X3_MIN_K and tiles >= X3_MIN_TILES) else _X3_FP32
    best, best_ms = _X3_FP32, None
    for cand in (_X3_FP32, 0, 1, 2) + ((3,) if tall else ()) + tuple(splitk):
        fn = run_fp32 if cand == _X3_FP32 else (lambda c=cand: run_split(c))
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        if best_ms is None or ms < best_ms:
            best, best_ms = cand, ms
    _x3_choice[key] = best
    return best


# The thin projection convolutions (N <= 32 output channels, K in {32, 96, 144}) of at least this many pixels go to the streaming
# kernel in the fp32 modes (exact fp32 products, so "f32" and "f32x3" alike): 5.0 TB/s against 3.8 TB/s for the tiled kernels.
THIN_PROJECT = True
THIN_MIN_ROWS = 65536


def linear_thin_supported(K: int, Nn: int) -> bool:
    return bool(N.load().icap_linear_thin_supported(K, Nn))


def linear_thin(x, w, bias=None, scale=None, residual=None, gate=None, hw=1, out=None):
    """y[M,N] = ((x[M,K] * gate[m // hw]) @ w[N,K]^T) * scale + bias (+ residual) for the thin projection layers (N <= 32,
    K in {32, 96, 144}): the streaming kernel with LDS-resident weights (icap_linear_thin_f32), exact fp32 products."""
    M, K = x.shape
    Nn = w.shape[0]
    if out is None:
        out = torch.empty((M, Nn), device=x.device, dtype=torch.float32)
    if gate is not None and (gate.shape[-1] != K or gate.shape[0] * hw != M):
        raise IcapError("linear_thin: gate must be [M // hw, K]")
    if residual is not None and residual.numel() != M * Nn:
        raise IcapError("linear_thin: residual has the wrong size")
    span = _Span("linear_thin", (M, K, Nn, residual is not None, gate is not None))
    N.check(N.load().icap_linear_thin_f32(_ck(x, "x"), _ck(w, "w"), _ck(scale, "scale"), _ck(bias, "bias"), _ck(residual, "residual"),
                                          _ck(gate, "gate"), hw, _ck(out, "out"), M, K, Nn, _stream()), "linear_thin")
    span.done()
    return out


# "f32x3" mode: layers without an activation function (the projection convolutions) of at least this many rows also offer the
# tuner the tall-activation kernel (linear_f32x3 cfg 3).
TALL_PROJECT = True
TALL_MIN_ROWS = 16384
SPLITK = True          # and layers of at most 256 64x64 tiles with K >= 1024 the split-K form


def linear_x3_auto(x, w, w3_cache: dict, key, bias=None, scale=None, act=ACT_NONE, residual=None, gate=None, hw=1):
    """One Linear / 1x1-convolution layer in "f32x3" mode: the split-product kernel or the fp32 kernel, whichever
    _x3_pick says; the weight's three bf16 planes are made on first use and kept in w3_cache[key]."""
    M, K = x.shape
    Nn = w.shape[0]

    def w3():
        t = w3_cache.get(key)
        if t is None:
            t = w3_cache[key] = split3(w.reshape(Nn, K).contiguous())
        return t

    def run_fp32():
        xx = x.view(M // hw, hw, 1, K) if gate is not None else x.view(M, 1, 1, K)
        rr = residual.view(xx.shape[0], xx.shape[1], 1, Nn) if residual is not None else None
        return conv2d(xx, w.reshape(Nn, 1, 1, K), scale, bias, residual=rr, gate=gate, act=act).view(M, Nn)

    def run_split(cfg):
        return linear_f32x3(x, w3(), bias, scale, act=act, residual=residual, gate=gate, hw=hw, cfg=cfg)

    tall = TALL_PROJECT and act == ACT_NONE and M >= TALL_MIN_ROWS and linear_f32x3_tall_supported(K, Nn, hw, gate is not None)
    # few output tiles and a long contraction: also offer the split-K form (2 / 4 / 8 slices of whole 32-wide steps)
    splitk = ()
    if SPLITK and K >= 1024 and ((M + 63) // 64) * ((Nn + 63) // 64) <= 256:
        splitk = tuple((ks << 4) | t for ks in (2, 4, 8) if K % (ks * 32) == 0 for t in (1, 2))
    c = _x3_pick(M, K, Nn, run_fp32, run_split, tall, splitk)
    return run_fp32() if c == _X3_FP32 else run_split(c)


def pointwise_conv(x, w, w16_cache: dict, key, scale=None, bias=None, residual=None, gate=None, act=ACT_NONE):
    """1x1 convolution x [B,H,W,Cin] with w [Cout,1,1,Cin]: the fp32 implicit-GEMM kernel, or - in bf16 compute mode and
    for shapes the bf16 kernel accepts - icap_linear_bf16_f32 with a cached bf16 copy of the weight."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    if COMPUTE_DTYPE == "bf16" and Cin % 8 == 0 and Cout % 4 == 0:
        w16 = w16_cache.get(key)
        if w16 is None:
            w16 = w16_cache[key] = to_bf16(w.reshape(Cout, Cin).contiguous())
        y = linear_bf16(x.view(B * H * W, Cin), w16, bias, scale, act=act,
                        residual=residual.view(B * H * W, Cout) if residual is not None else None, gate=gate, hw=H * W)
        return y.view(B, H, W, Cout)
    if x.dtype != torch.float32:
        raise IcapError("pointwise_conv: a bf16 activation needs the bf16 kernel (Cin % 8 == 0, Cout % 4 == 0, bf16 mode)")
    if (THIN_PROJECT and act == ACT_NONE and B * H * W >= THIN_MIN_ROWS and (gate is None or H * W >= 32)
            and x.is_contiguous() and linear_thin_supported(Cin, Cout)):
        y = linear_thin(x.view(B * H * W, Cin), w.reshape(Cout, Cin), bias, scale,
                        residual=residual.view(B * H * W, Cout) if residual is not None else None, gate=gate, hw=H * W)
        return y.view(B, H, W, Cout)
    if COMPUTE_DTYPE == "f32x3":
        y = linear_x3_auto(x.view(B * H * W, Cin), w, w16_cache, key + "_x3", bias, scale, act=act,
                           residual=residual.view(B * H * W, Cout) if residual is not None else None, gate=gate, hw=H * W)
        return y.view(B, H, W, Cout)
    return conv2d(x, w, scale, bias, residual=residual, gate=gate, act=act)


def dense_conv(x, w, w16_cache: dict, key, scale=None, bias=None, residual=None, *, stride=1, pad=(0, 0, 0, 0), act=ACT_NONE,
               out_bf16=False, cfg=-1):
    """KH x KW convolution (conv2d's arguments): the fp32 implicit-GEMM kernel, or - in bf16 compute mode and for
    Cin % 32 == 0 (or the 4-channel padded image), Cout % 4 == 0 - icap_conv2d_bf16_f32 with a cached bf16 copy of the
    weight.  In that mode x and residual may be bf16 tensors and out_bf16 asks for a bf16 result (activations kept in
    bf16 from layer to layer); the fp32 kernel takes and gives fp32 only."""
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    if not (COMPUTE_DTYPE == "bf16" and (Cin % 32 == 0 or Cin == 4) and Cout % 4 == 0):
        if out_bf16 or x.dtype != torch.float32 or (residual is not None and residual.dtype != torch.float32):
            raise IcapError("dense_conv: bf16 tensors need the bf16 kernel (bf16 mode, Cin % 32 == 0, Cout % 4 == 0)")
        return conv2d(x, w, scale, bias, residual=residual, stride=stride, pad=pad, act=act)
    w16 = w16_cache.get(key)
    if w16 is None:
        wk = w.contiguous()
        if Cin == 4:    # K = KH*KW*4 padded with zeros to a multiple of 32 (the kernel reads taps beyond KH*KW as zeros)
            k = KH * KW * 4
            wk = torch.nn.functional.pad(wk.view(Cout, k), (0, (k + 31) // 32 * 32 - k)).contiguous()
        w16 = w16_cache[key] = to_bf16(wk)
    pt, pb, pl, pr = pad
    Ho = conv_out_size(H, KH, stride, pt, pb)
    Wo = conv_out_size(W, KW, stride, pl, pr)
    x16 = x.dtype == torch.bfloat16
    r16 = residual is not None and residual.dtype == torch.bfloat16
    if x16 and Cin % 32:
        raise IcapError("dense_conv: a bf16 input needs Cin % 32 == 0")
    out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    if residual is not None and residual.numel() != out.numel():
        raise IcapError("dense_conv: residual has the wrong size")
    span = _Span("conv2d_bf16", (B, H, W, Cin, Cout, KH, stride, Ho, Wo, residual is not None, x16, bool(out_bf16)))
    N.check(N.load().icap_conv2d_bf16_f32(_ck(x, "x", x.dtype if x16 else torch.float32), _ck(w16, "w", torch.bfloat16), _ck(scale, "scale"),
                                          _ck(bias, "bias"), _ck(residual, "residual", torch.bfloat16 if r16 else torch.float32),
                                          _ck(out, "out", out.dtype), B, H, W, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo, act,
                                          int(x16) | (int(bool(out_bf16)) << 1) | (int(r16) << 2), cfg, _stream()), "conv2d_bf16")
    span.done()
    return out


def bf16_storage_ok(cin: int, cout: int) -> bool:
    """True when dense_conv can hand a bf16 [.., cout] result to the next dense layer / pooling (bf16 mode, channel counts
    the bf16 kernels take)."""
    return COMPUTE_DTYPE == "bf16" and (cin % 32 == 0 or cin == 4) and cout % 32 == 0


def bf16_activations_ok(cin: int, cout: int) -> bool:
    """True when a [.., cin] activation may be handed to pointwise_conv as bf16 (bf16 mode and a shape its kernel takes)."""
    return COMPUTE_DTYPE == "bf16" and cin % 8 == 0 and cout % 4 == 0


# Raw-pointer writes (optimizer kernels, BatchNorm running statistics, graph replays) change weights without changing
# the tensors' (data_ptr, _version), so the eval-mode caches of kernel-layout weight copies (backbones.PreparedCache)
# also key on this counter: everything that writes a parameter or buffer through the library advances it.
_weights_generation = 0


def bump_weights_generation() -> None:
    global _weights_generation
    _weights_generation += 1


def weights_generation() -> int:
    return _weights_generation


def set_autotune(flag: bool) -> None:
    global AUTOTUNE
    AUTOTUNE = bool(flag)


def _conv_entry(key):
    """icap_conv2d_f32, or the tuning entry point the first time `key` is seen with autotune on."""
    lib = N.load()
    if AUTOTUNE and key not in _tuned_shapes:
        _tuned_shapes.add(key)
        return lib.icap_conv2d_tune_f32
    return lib.icap_conv2d_f32


# Optional launch tracer for measurement harnesses: a callable (op_name, signature) -> None or a pair of
# torch.cuda.Event that are recorded on the launch stream immediately before and after the kernel.
_trace = None


def set_trace(fn) -> None:
    global _trace
    _trace = fn


class _Span:
    __slots__ = ("ev",)

    def __init__(self, name, sig):
        self.ev = _trace(name, sig) if _trace is not None else None
        if self.ev is not None:
            self.ev[0].record()

    def done(self):
        if self.ev is not None:
            self.ev[1].record()


def _ck(t: torch.Tensor | None, name: str, dtype=torch.float32):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise IcapError(f"{name}: expected a tensor on the HIP device, got {getattr(t, 'device', type(t))}; "
                        "the MI355X path has no CPU fallback")
    if t.dtype != dtype:
        raise IcapError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise IcapError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def conv_out_size(size: int, k: int, stride: int, pad_lo: int, pad_hi: int) -> int:
    return (size + pad_lo + pad_hi - k) // stride + 1


def nchw_to_nhwc_pad(images: torch.Tensor, cp: int = 4) -> torch.Tensor:
    B, Cc, H, W = images.shape
    out = torch.empty((B, H, W, cp), device=images.device, dtype=torch.float32)
    N.check(N.load().icap_nchw_to_nhwc_pad_f32(_ck(images, "images"), _ck(out, "out"), B, Cc, H, W, cp, _stream()),
            "nchw_to_nhwc_pad")
    return out


class untuned:
    """`with ops.untuned(): ...` - conv2d / linear run on the shape heuristic's tile configuration, whatever the tuner
    remembered (verification reruns)."""

    def __enter__(self):
        global AUTOTUNE, THIN_PROJECT, _X3_USE_TUNED
        self.prev, self.tune, self.thin, self.x3 = N.load().icap_conv2d_use_tuned(0), AUTOTUNE, THIN_PROJECT, _X3_USE_TUNED
        AUTOTUNE = False
        THIN_PROJECT = False         # the rerun goes through the general implicit-GEMM kernel, not the specialised streaming one
        _X3_USE_TUNED = False        # and the split-product layers on the shape rule, never on a remembered (tall) configuration
        return self

    def __exit__(self, *exc):
        global AUTOTUNE, THIN_PROJECT, _X3_USE_TUNED
        N.load().icap_conv2d_use_tuned(self.prev)
        AUTOTUNE, THIN_PROJECT, _X3_USE_TUNED = self.tune, self.thin, self.x3
        return False


def conv2d_num_configs() -> int:
    return N.load().icap_conv2d_num_configs()


def conv2d(x, w, scale=None, bias=None, residual=None, gate=None, *, stride=1, pad=(0, 0, 0, 0), act=ACT_NONE, out=None, cfg=None):
    """x [B,H,W,Cin] NHWC, w [Cout,KH,KW,Cin]; pad = (top, bottom, left, right).  cfg: run that tile configuration
    (0 <= cfg < conv2d_num_configs()) instead of the heuristic / tuned one; raises when it cannot run the shape."""
    B, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    if Cin2 != Cin:
        raise IcapError(f"conv2d: weight expects {Cin2} input channels, activation has {Cin}")
    pt, pb, pl, pr = pad
    Ho = conv_out_size(H, KH, stride, pt, pb)
    Wo = conv_out_size(W, KW, stride, pl, pr)
    if out is None:
        out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    elif out.numel() != B * Ho * Wo * Cout:
        raise IcapError("conv2d: out has the wrong size")
    if residual is not None and residual.numel() != out.numel():
        raise IcapError("conv2d: residual has the wrong size")
    if gate is not None and tuple(gate.shape) != (B, Cin):
        raise IcapError("conv2d: gate must be [B, Cin]")
    for v, nm in ((scale, "scale"), (bias, "bias")):
        if v is not None and v.numel() != Cout:
            raise IcapError(f"conv2d: {nm} must have Cout={Cout} elements")
    if cfg is not None:
        N.check(N.load().icap_conv2d_cfg_f32(
            _ck(x, "x"), _ck(w, "w"), _ck(scale, "scale"), _ck(bias, "bias"), _ck(residual, "residual"),
            _ck(gate, "gate"), _ck(out, "out"), B, H, W, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo, act, int(cfg), _stream()),
            f"conv2d[cfg {cfg}]")
        return out
    span = _Span("conv2d", (B, H, W, Cin, Cout, KH, stride, Ho, Wo, residual is not None, gate is not None))
    N.check(_conv_entry((B, H, W, Cin, Cout, KH, KW, stride, Ho, Wo))(
        _ck(x, "x"), _ck(w, "w"), _ck(scale, "scale"), _ck(bias, "bias"), _ck(residual, "residual"),
        _ck(gate, "gate"), _ck(out, "out"), B, H, W, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo, act, _stream()), "conv2d")
    span.done()
    return out


def linear(x, w, bias=None, scale=None, act=ACT_NONE, residual=None, out=None):
    """y[M,N] = act((x[M,K] @ w[N,K]^T) * scale + bias (+ residual)) on the matrix cores."""
    M, K = x.shape
    Nn, K2 = w.shape
    if K != K2:
        raise IcapError(f"linear: x has K={K}, w has K={K2}")
    if out is None:
        out = torch.empty((M, Nn), device=x.device, dtype=torch.float32)
    if residual is not None and residual.numel() != M * Nn:
        raise IcapError("linear: residual has the wrong size")
    if TRAIN_DTYPE == "bf16" and K % 8 == 0 and Nn % 4 == 0 and M >= TRAIN_BF16_MIN_ROWS:
        # bf16 training step: same epilogue, bf16 products (the weight's bf16 copy is made once per step)
        w16 = step_cached("bf16", w, lambda: to_bf16(w))
        return linear_bf16(x, w16, bias, scale, act=act, residual=residual, out=out)
    if TRAIN_DTYPE == "f32x3" and K % 8 == 0 and Nn % 4 == 0 and M >= TRAIN_BF16_MIN_ROWS:
        def run_fp32(o=None):
            return _linear_fp32(x, w, bias, scale, act, residual, torch.empty_like(out) if o is None else o)

        def run_split(c, o=None):
            return linear_f32x3(x, step_cached("split3", w, lambda: split3(w.contiguous())), bias, scale, act=act, residual=residual,
                                out=torch.empty_like(out) if o is None else o, cfg=c)

        c = _x3_pick(M, K, Nn, run_fp32, run_split)      # the tuner's trial runs write to scratch results (out may alias residual)
        return run_fp32(out) if c == _X3_FP32 else run_split(c, out)
    return _linear_fp32(x, w, bias, scale, act, residual, out)


class StatParts:
    """Column statistics a producing kernel left for the train-mode BatchNorm that follows (train_ops.batchnorm_train):
    part [P, 2, C] (sum, squared deviations per block of rows), cnt [P] values per block or None when block p holds
    min(rows_per_part, M - p * rows_per_part) rows."""
    __slots__ = ("part", "cnt", "P", "rows_per_part")

    def __init__(self, part, cnt, P, rows_per_part):
        self.part, self.cnt, self.P, self.rows_per_part = part, cnt, P, rows_per_part


def _conv_stats(x4, w4, stride, pad, Ho, Wo):
    """Plain convolution on the fp32 implicit-GEMM kernel + the column statistics of its result (icap_conv2d_stats_f32)."""
    lib = N.load()
    B, H, W, Cin = x4.shape
    Cout, KH, KW, _ = w4.shape
    pt, pb, pl, pr = pad
    M = B * Ho * Wo
    y = torch.empty((B, Ho, Wo, Cout), device=x4.device, dtype=torch.float32)
    key = (B, H, W, Cin, Cout, KH, KW, stride, Ho, Wo)
    if AUTOTUNE and key not in _tuned_shapes:
        # first sight of the shape with tuning on: the tuning entry point measures the tile configurations (and computes y);
        # the statistics launch below then runs the winner
        _tuned_shapes.add(key)
        N.check(lib.icap_conv2d_tune_f32(_ck(x4, "x"), _ck(w4, "w"), None, None, None, None, _ck(y, "y"), B, H, W, Cin, Cout, KH, KW,
                                         stride, pt, pl, Ho, Wo, ACT_NONE, _stream()), "conv2d_tune")
    cap = lib.icap_conv2d_stats_max_parts(M)
    part = torch.empty((cap, 2, Cout), device=x4.device, dtype=torch.float32)
    info = (C.c_int * 2)()
    span = _Span("linear" if KH == 1 and H == 1 else "conv2d", (M, KH * KW * Cin, Cout, False) if KH == 1 and H == 1 else
                 (B, H, W, Cin, Cout, KH, stride, Ho, Wo, False, False))
    N.check(lib.icap_conv2d_stats_f32(_ck(x4, "x"), _ck(w4, "w"), _ck(y, "y"), B, H, W, Cin, Cout, KH, KW, stride, pt, pl, Ho, Wo,
                                      part.data_ptr(), part.numel(), C.cast(info, C.c_void_p), _stream()), "conv2d_stats")
    span.done()
    return y, StatParts(part, None, int(info[0]), int(info[1]))


def conv2d_stats(x, w, *, stride=1, pad=(0, 0, 0, 0)):
    """x [B,H,W,Cin], w [Cout,KH,KW,Cin] -> (y, StatParts): a convolution in front of a train-mode BatchNorm."""
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    Ho = conv_out_size(H, KH, stride, pad[0], pad[1])
    Wo = conv_out_size(W, KW, stride, pad[2], pad[3])
    if Cout % 4:
        return conv2d(x, w, stride=stride, pad=pad), None
    return _conv_stats(x, w, stride, pad, Ho, Wo)


def linear_stats(x, w):
    """y = x @ w^T plus the column statistics of y for the train-mode BatchNorm that follows -> (y, StatParts or None).
    Same kernel choice as linear(); None when the chosen kernel has no statistics epilogue (the caller then takes the
    statistics from y)."""
    M, K = x.shape
    Nn = w.shape[0]
    if Nn % 4 or TRAIN_DTYPE == "bf16":
        return linear(x, w), None
    if TRAIN_DTYPE == "f32x3" and K % 8 == 0 and M >= TRAIN_BF16_MIN_ROWS:
        def w3():
            return step_cached("split3", w, lambda: split3(w.contiguous()))

        def run_fp32():
            return _linear_fp32(x, w, None, None, ACT_NONE, None, torch.empty((M, Nn), device=x.device, dtype=torch.float32))

        def run_split(c):
            return linear_f32x3(x, w3(), cfg=c)

        c = _x3_pick(M, K, Nn, run_fp32, run_split)
        if c != _X3_FP32:
            lib = N.load()
            y = torch.empty((M, Nn), device=x.device, dtype=torch.float32)
            part = torch.empty(((M + 63) // 64 + 2, 2, Nn), device=x.device, dtype=torch.float32)
            info = (C.c_int * 2)()
            span = _Span("linear_f32x3", (M, K, Nn, False, False))
            N.check(lib.icap_linear_f32x3_stats_f32(_ck(x, "x"), w3().data_ptr(), _ck(y, "y"), M, K, Nn, c, part.data_ptr(), part.numel(),
                                                    C.cast(info, C.c_void_p), _stream()), "linear_f32x3_stats")
            span.done()
            return y, StatParts(part, None, int(info[0]), int(info[1]))
    y, st = _conv_stats(x.view(M, 1, 1, K), w.view(Nn, 1, 1, K), 1, (0, 0, 0, 0), 1, 1)
    return y.view(M, Nn), st


def dwconv2d_stats(x, w, *, ks, stride=1, pad=(0, 0, 0, 0)):
    """Plain depthwise convolution + the per-workgroup statistics of its result -> (y [B,Ho,Wo,C], StatParts)."""
    B, H, W, Cc = x.shape
    pt, pb, pl, pr = pad
    Ho = conv_out_size(H, ks, stride, pt, pb)
    Wo = conv_out_size(W, ks, stride, pl, pr)
    if tuple(w.shape) != (ks * ks, Cc):
        raise IcapError("dwconv2d: weight must be [ks*ks, C]")
    lib = N.load()
    y = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    P = B * lib.icap_dwconv_chunks(Ho, Wo, Cc, ks, stride)
    part = torch.empty((P, 2, Cc), device=x.device, dtype=torch.float32)
    cnt = torch.empty(P, device=x.device, dtype=torch.float32)
    span = _Span("dwconv2d", (B, H, W, Cc, ks, stride, Ho, Wo))
    N.check(lib.icap_dwconv2d_stats_f32(_ck(x, "x"), _ck(w, "w"), _ck(y, "y"), B, H, W, Cc, ks, stride, pt, pl, Ho, Wo,
                                        part.data_ptr(), cnt.data_ptr(), _stream()), "dwconv2d_stats")
    span.done()
    return y, StatParts(part, cnt, P, 0)


def _linear_fp32(x, w, bias, scale, act, residual, out):
    M, K = x.shape
    Nn = w.shape[0]
    span = _Span("linear", (M, K, Nn, residual is not None))
    N.check(_conv_entry((M, 1, 1, K, Nn, 1, 1, 1, 1, 1))(
        _ck(x, "x"), _ck(w, "w"), _ck(scale, "scale"), _ck(bias, "bias"), _ck(residual, "residual"), None,
        _ck(out, "out"), M, 1, 1, K, Nn, 1, 1, 1, 0, 0, 1, 1, act, _stream()), "linear")
    span.done()
    return out


def dwconv2d(x, w, scale=None, bias=None, *, ks, stride=1, pad=(0, 0, 0, 0), act=ACT_NONE, want_partial=False):
    """x [B,H,W,C], w [ks*ks, C] -> (y, partial or None)."""
    B, H, W, Cc = x.shape
    pt, pb, pl, pr = pad
    Ho = conv_out_size(H, ks, stride, pt, pb)
    Wo = conv_out_size(W, ks, stride, pl, pr)
    if tuple(w.shape) != (ks * ks, Cc):
        raise IcapError("dwconv2d: weight must be [ks*ks, C]")
    y = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    lib = N.load()
    partial = None
    if want_partial:
        partial = torch.empty((B, lib.icap_dwconv_chunks(Ho, Wo, Cc, ks, stride), Cc), device=x.device, dtype=torch.float32)
    span = _Span("dwconv2d", (B, H, W, Cc, ks, stride, Ho, Wo))
    N.check(lib.icap_dwconv2d_f32(_ck(x, "x"), _ck(w, "w"), _ck(scale, "scale"), _ck(bias, "bias"), _ck(y, "y"),
                                  _ck(partial, "partial"), B, H, W, Cc, ks, stride, pt, pl, Ho, Wo, act, _stream()),
            "dwconv2d")
    span.done()
    return y, partial


def expand_dwconv_supported(H, W, Cin, Cexp, ks, stride, pad) -> bool:
    """True when the fused expand + depthwise kernel covers this layer shape."""
    pt, pb, pl, pr = pad
    lib = N.load()
    if _img_x3_s2(lib, H, W, Cin, Cexp, ks, stride, pad):
        return True          # the stride-2 form of the whole-image split-product kernel (14 x 14 -> 7 x 7)
    if lib.icap_expand_dwconv_whole_image(H, W, Cin, Cexp, ks, stride) and (COMPUTE_DTYPE == "bf16" or (pt, pl) != (ks // 2, ks // 2)):
        return False         # the whole-image kernel of the 14x14 / 7x7 blocks: fp32 tensors, 'same' padding
    return lib.icap_expand_dwconv_chunks(conv_out_size(H, ks, stride, pt, pb), conv_out_size(W, ks, stride, pl, pr),
                                         Cin, Cexp, ks, stride) > 0


def _img_x3_s2(lib, H, W, Cin, Cexp, ks, stride, pad) -> bool:
    pt, pb, pl, pr = pad
    return bool(COMPUTE_DTYPE == "f32x3" and IMG_X3 and IMG_X3_S2 and stride == 2
                and conv_out_size(H, ks, stride, pt, pb) == 7 and conv_out_size(W, ks, stride, pl, pr) == 7
                and lib.icap_expand_dwconv_img_x3_s2_supported(H, W, Cin, Cexp, ks, pt, pl))


IMG_X3_S2 = True   # ... and the 14 x 14 -> 7 x 7 block on its stride-2 form (False: expand GEMM + depthwise kernel)
IMG_X3 = True      # "f32x3" mode: the whole-image layer shapes on icap_expand_dwconv_img_x3_f32 (False: the fp32 whole-image kernel)


def expand_dwconv2d(x, w_exp, exp_scale, exp_bias, w_dw, dw_scale, dw_bias, *, ks, stride=1, pad=(0, 0, 0, 0), want_partial=False,
                    out_bf16=False, cache=None, key="expand_x3p"):
    """x [B,H,W,Cin], w_exp [Cexp,1,1,Cin] or [Cexp,Cin], w_dw [ks*ks, Cexp] -> (swish(bn(dw(swish(bn(expand(x)))))), partial).
    cache / key: where the packed split weights of the whole-image split-product kernel are kept between calls (a dict
    that lives as long as the weight, e.g. the layer's prepared-weights entry); without it they are packed per call."""
    B, H, W, Cin = x.shape
    Cexp = w_exp.shape[0]
    pt, pb, pl, pr = pad
    Ho = conv_out_size(H, ks, stride, pt, pb)
    Wo = conv_out_size(W, ks, stride, pl, pr)
    if w_exp.numel() != Cexp * Cin or tuple(w_dw.shape) != (ks * ks, Cexp):
        raise IcapError("expand_dwconv2d: weights must be [Cexp, Cin] and [ks*ks, Cexp]")
    lib = N.load()
    s2 = _img_x3_s2(lib, H, W, Cin, Cexp, ks, stride, pad)
    nch = 1 if s2 else lib.icap_expand_dwconv_chunks(Ho, Wo, Cin, Cexp, ks, stride)
    if nch <= 0:
        raise IcapError(f"expand_dwconv2d: no fused kernel for Cin={Cin}, {ks}x{ks} stride {stride}")
    bf16 = COMPUTE_DTYPE == "bf16"
    if out_bf16 and not bf16:
        raise IcapError("expand_dwconv2d: a bf16 result needs the bf16 compute mode")
    ydt = torch.bfloat16 if out_bf16 else torch.float32
    y = torch.empty((B, Ho, Wo, Cexp), device=x.device, dtype=ydt)
    partial = torch.empty((B, nch, Cexp), device=x.device, dtype=torch.float32) if want_partial else None
    span = _Span("expand_dwconv2d", (B, H, W, Cin, Cexp, ks, stride, Ho, Wo, out_bf16))
    head = (_ck(x, "x"), _ck(w_exp, "w_exp"), _ck(exp_scale, "exp_scale"), _ck(exp_bias, "exp_bias"), _ck(w_dw, "w_dw"),
            _ck(dw_scale, "dw_scale"), _ck(dw_bias, "dw_bias"))
    tail = (_ck(partial, "partial"), B, H, W, Cin, Cexp, ks, stride, pt, pl, Ho, Wo, _stream())
    if bf16:
        N.check(lib.icap_expand_dwconv_bf16_f32(*head, _ck(y, "y", ydt), int(out_bf16), *tail), "expand_dwconv2d")
    elif s2 or (COMPUTE_DTYPE == "f32x3" and IMG_X3 and stride == 1 and (pt, pl) == (ks // 2, ks // 2)
                and lib.icap_expand_dwconv_img_x3_we_bytes(H, W, Cin, Cexp, ks) > 0):
        wp = cache.get(key) if cache is not None else None
        if wp is None:
            wp = torch.empty(lib.icap_expand_dwconv_img_x3_we_bytes(H, W, Cin, Cexp, ks) // 2, device=x.device, dtype=torch.bfloat16)
            N.check(lib.icap_expand_dwconv_img_x3_pack(_ck(w_exp, "w_exp"), wp.data_ptr(), H, W, Cin, Cexp, ks, _stream()), "expand_dwconv2d pack")
            if cache is not None:
                cache[key] = wp
        run = lib.icap_expand_dwconv_img_x3_s2_f32 if s2 else lib.icap_expand_dwconv_img_x3_f32
        N.check(run(head[0], wp.data_ptr(), *head[2:], _ck(y, "y"), _ck(partial, "partial"), B, H, W, Cin, Cexp, ks, pt, pl, _stream()),
                "expand_dwconv2d")
    elif COMPUTE_DTYPE == "f32x3":
        N.check(lib.icap_expand_dwconv_f32x3_f32(*head, _ck(y, "y"), *tail), "expand_dwconv2d")
    else:
        N.check(lib.icap_expand_dwconv_f32(*head, _ck(y, "y"), *tail), "expand_dwconv2d")
    span.done()
    return y, partial


STEM_NCHW = True       # fp32 modes: the fused stem kernel reads the planar NCHW image directly (False: interleave + pad pass first)
STEM_VALU = True       # ... and, given the k-major stem filter (w_stem_k), runs on the vector-ALU kernel (icap_stem_dwconv_valu_f32)


def stem_weight_k(w_stem4: torch.Tensor) -> torch.Tensor:
    """[C,3,3,4] kernel-layout stem filter (fourth input channel zero) -> [27, C], k = (kh * 3 + kw) * 3 + input channel."""
    Cc = w_stem4.shape[0]
    return w_stem4[..., :3].reshape(Cc, 27).t().contiguous()


def stem_dwconv2d(x, w_stem, stem_scale, stem_bias, w_dw, dw_scale, dw_bias, *, stem_pad=(0, 1, 0, 1), pad=(1, 1, 1, 1), want_partial=False,
                  out_bf16=False, nchw=False, w_stem_k=None):
    """x [B,H,W,4] (image, channels padded to 4) - or, with nchw=True, the planar fp32 image [B,3,H,W] as the reference hands it
    over (fp32 modes only) -, w_stem [C,3,3,4], w_dw [9, C] ->
    (swish(bn(dw3x3(swish(bn(conv3x3 stride 2 (x)))))), partial): EfficientNet stem + block 0's depthwise convolution."""
    if nchw:
        B, c4, H, W = x.shape
        c4 += 1
    else:
        B, H, W, c4 = x.shape
    Cc = w_stem.shape[0]
    if c4 != 4 or tuple(w_stem.shape[1:]) != (3, 3, 4) or tuple(w_dw.shape) != (9, Cc):
        raise IcapError("stem_dwconv2d: x must be [B,H,W,4] (or [B,3,H,W] with nchw=True), w_stem [C,3,3,4], w_dw [9,C]")
    Hs = conv_out_size(H, 3, 2, stem_pad[0], stem_pad[1])
    Ws = conv_out_size(W, 3, 2, stem_pad[2], stem_pad[3])
    Ho = conv_out_size(Hs, 3, 1, pad[0], pad[1])
    Wo = conv_out_size(Ws, 3, 1, pad[2], pad[3])
    lib = N.load()
    bf16 = COMPUTE_DTYPE == "bf16"
    if out_bf16 and not bf16:
        raise IcapError("stem_dwconv2d: a bf16 result needs the bf16 compute mode")
    ydt = torch.bfloat16 if out_bf16 else torch.float32
    y = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=ydt)
    partial = torch.empty((B, lib.icap_stem_dwconv_chunks(Ho, Wo, Cc), Cc), device=x.device, dtype=torch.float32) if want_partial else None
    span = _Span("stem_dwconv2d", (B, H, W, Cc, Hs, Ws, Ho, Wo, out_bf16, bool(nchw)))
    head = (_ck(x, "x"), _ck(w_stem, "w_stem"), _ck(stem_scale, "stem_scale"), _ck(stem_bias, "stem_bias"), _ck(w_dw, "w_dw"),
            _ck(dw_scale, "dw_scale"), _ck(dw_bias, "dw_bias"))
    tail = (_ck(partial, "partial"), B, H, W, Hs, Ws, Cc, stem_pad[0], stem_pad[2], pad[0], pad[2], Ho, Wo, _stream())
    if nchw and bf16:
        raise IcapError("stem_dwconv2d: the planar image is taken in the fp32 modes only")
    if bf16:
        N.check(lib.icap_stem_dwconv_bf16_f32(*head, _ck(y, "y", ydt), int(out_bf16), *tail), "stem_dwconv2d")
    elif nchw and STEM_VALU and w_stem_k is not None and lib.icap_stem_dwconv_valu_supported(Cc):
        if tuple(w_stem_k.shape) != (27, Cc):
            raise IcapError("stem_dwconv2d: w_stem_k must be [27, C]")
        N.check(lib.icap_stem_dwconv_valu_f32(head[0], _ck(w_stem_k, "w_stem_k"), *head[2:], _ck(y, "y"), *tail), "stem_dwconv2d")
    elif nchw:
        N.check(lib.icap_stem_dwconv_nchw_f32(*head, _ck(y, "y"), *tail), "stem_dwconv2d")
    else:
        N.check(lib.icap_stem_dwconv_f32(*head, _ck(y, "y"), *tail), "stem_dwconv2d")
    span.done()
    return y, partial


def se_gate(partial, hw: int, wr, br, we, be):
    B, nchunks, Cc = partial.shape
    S = wr.shape[0]
    if tuple(wr.shape) != (S, Cc) or tuple(we.shape) != (S, Cc) or br.numel() != S or be.numel() != Cc:
        raise IcapError("se_gate: weight shapes do not match")
    gate = torch.empty((B, Cc), device=partial.device, dtype=torch.float32)
    N.check(N.load().icap_se_gate_f32(_ck(partial, "partial"), nchunks, 1.0 / hw, _ck(wr, "wr"), _ck(br, "br"),
                                      _ck(we, "we"), _ck(be, "be"), _ck(gate, "gate"), B, Cc, S, _stream()), "se_gate")
    return gate


def maxpool2d(x, ks, stride, pad):
    B, H, W, Cc = x.shape
    Ho = conv_out_size(H, ks, stride, pad, pad)
    Wo = conv_out_size(W, ks, stride, pad, pad)
    if x.dtype == torch.bfloat16:       # activations kept in bf16 between the dense layers (bf16 compute mode)
        y = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=torch.bfloat16)
        N.check(N.load().icap_maxpool2d_bf16(_ck(x, "x", torch.bfloat16), _ck(y, "y", torch.bfloat16), B, H, W, Cc, ks, stride, pad, Ho, Wo,
                                             _stream()), "maxpool2d")
        return y
    y = torch.empty((B, Ho, Wo, Cc), device=x.device, dtype=torch.float32)
    N.check(N.load().icap_maxpool2d_f32(_ck(x, "x"), _ck(y, "y"), B, H, W, Cc, ks, stride, pad, Ho, Wo, _stream()), "maxpool2d")
    return y


def adaptive_avgpool2d(x, oh, ow):
    B, H, W, Cc = x.shape
    y = torch.empty((B, oh, ow, Cc), device=x.device, dtype=torch.float32)
    if x.dtype == torch.bfloat16:       # bf16 in, fp32 out
        N.check(N.load().icap_adaptive_avgpool2d_bf16_f32(_ck(x, "x", torch.bfloat16), _ck(y, "y"), B, H, W, Cc, oh, ow, _stream()),
                "adaptive_avgpool2d")
        return y
    N.check(N.load().icap_adaptive_avgpool2d_f32(_ck(x, "x"), _ck(y, "y"), B, H, W, Cc, oh, ow, _stream()), "adaptive_avgpool2d")
    return y


def embedding(table, idx, add=None):
    V, E = table.shape
    rows = idx.numel()
    out = torch.empty((rows, E), device=table.device, dtype=torch.float32)
    err = torch.zeros(1, device=table.device, dtype=torch.int32)
    if add is not None and add.numel() != rows * E:
        raise IcapError("embedding: add must be [rows, E]")
    N.check(N.load().icap_embedding_f32(_ck(table, "table"), _ck(idx, "idx", torch.int64), _ck(add, "add"), _ck(out, "out"),
                                        rows, E, V, _ck(err, "err", torch.int32), _stream()), "embedding")
    return out, err


def layernorm(x, gamma, beta, res=None, eps=1e-5, pre_scale=1.0, split=False):
    """split=True: returns split3(LayerNorm(...)) - the bf16 [3, rows, E] planes a split-product GEMM takes as a pre-split
    activation - written by the LayerNorm kernel itself; the fp32 result is not stored."""
    rows, E = x.shape
    res_rows = 1
    if res is not None:
        if res.shape[-1] != E or rows % (res.numel() // E):
            raise IcapError("layernorm: residual rows must divide x rows")
        res_rows = res.numel() // E
    if split:
        y3 = torch.empty((3, rows, E), device=x.device, dtype=torch.bfloat16)
        N.check(N.load().icap_layernorm_split3_f32(_ck(x, "x"), _ck(res, "res"), res_rows, _ck(gamma, "gamma"), _ck(beta, "beta"),
                                                   None, y3.data_ptr(), rows, E, eps, pre_scale, _stream()), "layernorm_split3")
        return y3
    y = torch.empty_like(x)
    N.check(N.load().icap_layernorm_f32(_ck(x, "x"), _ck(res, "res"), res_rows, _ck(gamma, "gamma"), _ck(beta, "beta"),
                                        _ck(y, "y"), rows, E, eps, pre_scale, _stream()), "layernorm")
    return y


def argmax_rows(logits, want_gap=False):
    rows, V = logits.shape
    idx = torch.empty(rows, device=logits.device, dtype=torch.int64)
    gap = torch.empty(rows, device=logits.device, dtype=torch.float32) if want_gap else None
    N.check(N.load().icap_argmax_rows_f32(_ck(logits, "logits"), V, _ck(idx, "idx", torch.int64), 1, rows, V,
                                          _ck(gap, "gap"), _stream()), "argmax_rows")
    return (idx, gap) if want_gap else idx


def to_bf16(w: torch.Tensor) -> torch.Tensor:
    """fp32 tensor -> bf16 copy (round to nearest even), same shape."""
    n = w.numel()
    if n % 2:
        raise IcapError("to_bf16: element count must be even")
    out = torch.empty(w.shape, device=w.device, dtype=torch.bfloat16)
    N.check(N.load().icap_to_bf16_f32(_ck(w, "w"), out.data_ptr(), n, _stream()), "to_bf16")
    return out


def split3(w: torch.Tensor) -> torch.Tensor:
    """fp32 tensor [..] -> bf16 [3, ..]: the exact three-way bf16 split (hi, mid, lo planes) linear_f32x3 multiplies with."""
    n = w.numel()
    if n % 2:
        raise IcapError("split3: element count must be even")
    out = torch.empty((3,) + tuple(w.shape), device=w.device, dtype=torch.bfloat16)
    N.check(N.load().icap_split3_f32(_ck(w, "w"), out.data_ptr(), n, _stream()), "split3")
    return out


_splitk_ws: dict = {}     # device -> the workspace of the split-K launches (launches on one stream run in order, so one is enough)


def _x_planes(x):
    """(M, K, pre-split?) of an activation operand: an fp32 [M, K] matrix or its split3 planes, bf16 [3, M, K]."""
    if x.dtype == torch.bfloat16:
        if x.dim() != 3 or x.shape[0] != 3:
            raise IcapError("linear_f32x3: bf16 activations must be split3(x) planes [3, M, K]")
        return x.shape[1], x.shape[2], True
    return x.shape[0], x.shape[1], False


def linear_f32x3(x, w3, bias=None, scale=None, act=ACT_NONE, residual=None, gate=None, hw=1, out=None, cfg=-1):
    """y[M,N] = act((x[M,K] (* gate[m // hw]) @ w[N,K]^T) * scale + bias (+ residual)), fp32-accurate, on the bf16 matrix
    cores: six exact bf16 products per multiply from three-way splits of both operands (w3 = split3(w); x fp32 and split
    on the fly, or split3(x) planes - no gate then), fp32 accumulation."""
    M, K, pre = _x_planes(x)
    if w3.dtype != torch.bfloat16 or w3.dim() < 2 or w3.shape[0] != 3 or w3.numel() // (3 * w3.shape[1]) != K:
        raise IcapError(f"linear_f32x3: w3 must be split3(w) of a [N, {K}] weight")
    Nn = w3.shape[1]
    if out is None:
        out = torch.empty((M, Nn), device=x.device, dtype=torch.float32)
    if residual is not None and residual.numel() != M * Nn:
        raise IcapError("linear_f32x3: residual has the wrong size")
    if gate is not None and (pre or gate.shape[-1] != K or gate.shape[0] * hw != M):
        raise IcapError("linear_f32x3: gate must be [M // hw, K] (and x fp32)")
    span = _Span("linear_f32x3", (M, K, Nn, residual is not None, gate is not None))
    if cfg >= 16:       # split-K: cfg = (slices << 4) | tile configuration; partial products go through a cached workspace
        if pre:
            raise IcapError("linear_f32x3: the split-K form takes fp32 activations")
        ks = cfg >> 4
        lib = N.load()
        need = lib.icap_linear_f32x3_splitk_ws_floats(M, Nn, ks)
        ws = _splitk_ws.get(x.device)
        if ws is None or ws.numel() < need:
            ws = _splitk_ws[x.device] = torch.empty(need, device=x.device, dtype=torch.float32)
        N.check(lib.icap_linear_f32x3_splitk_f32(_ck(x, "x"), _ck(w3, "w3", torch.bfloat16), _ck(scale, "scale"), _ck(bias, "bias"),
                                                 _ck(residual, "residual"), _ck(gate, "gate"), hw, _ck(out, "out"), ws.data_ptr(), ws.numel(),
                                                 M, K, Nn, act, cfg & 15, ks, _stream()), "linear_f32x3_splitk")
    else:
        N.check(N.load().icap_linear_f32x3_f32(_ck(x, "x", x.dtype), int(pre), _ck(w3, "w3", torch.bfloat16), _ck(scale, "scale"),
                                               _ck(bias, "bias"), _ck(residual, "residual"), _ck(gate, "gate"), hw, _ck(out, "out"), M, K,
                                               Nn, act, cfg, _stream()), "linear_f32x3")
    span.done()
    return out


def linear_f32x3_argmax(x, w3, bias=None, scale=None, want_logits=False):
    """fp32-accurate vocabulary projection (split bf16 products) with fused arg-max partials per 128-column tile; x fp32
    [M, K] or its split3 planes."""
    M, K, pre = _x_planes(x)
    Nn = w3.shape[1]
    lib = N.load()
    P = lib.icap_linear_bf16_argmax_parts(Nn)
    pv = torch.empty((M, P), device=x.device, dtype=torch.float32)
    pi = torch.empty((M, P), device=x.device, dtype=torch.int32)
    logits = torch.empty((M, Nn), device=x.device, dtype=torch.float32) if want_logits else None
    span = _Span("linear_f32x3", (M, K, Nn, False, False))
    N.check(lib.icap_linear_f32x3_argmax_f32(_ck(x, "x", x.dtype), int(pre), _ck(w3, "w3", torch.bfloat16), _ck(scale, "scale"),
                                             _ck(bias, "bias"), _ck(logits, "logits"), _ck(pv, "pv"), _ck(pi, "pi", torch.int32), M, K, Nn,
                                             _stream()), "linear_f32x3_argmax")
    span.done()
    return pv, pi, logits


def linear_x3dl(x3, w3, bias=None, scale=None, act=ACT_NONE, residual=None, want_f32=True, want_split=False, cfg=-1):
    """fp32-accurate y = act((x @ w^T) * scale + bias (+ residual)) from PRE-SPLIT operands streamed straight into LDS
    (icap_linear_x3dl_f32): x3 = split3(x) [3, M, K], w3 = split3(w) [3, N, K], K % 32 == 0.  Returns (y fp32 or None,
    y3 = split3(y) planes or None) - y3 feeds the next such GEMM without a separate split pass."""
    if x3.dtype != torch.bfloat16 or x3.dim() != 3 or x3.shape[0] != 3 or w3.dtype != torch.bfloat16 or w3.shape[0] != 3:
        raise IcapError("linear_x3dl: x3 and w3 must be split3() planes")
    M, K = x3.shape[1], x3.shape[2]
    Nn = w3.shape[1]
    if w3.numel() // (3 * Nn) != K:
        raise IcapError(f"linear_x3dl: x has K={K}, w3 is split3 of a [{Nn}, {w3.numel() // (3 * Nn)}] weight")
    y = torch.empty((M, Nn), device=x3.device, dtype=torch.float32) if want_f32 else None
    y3 = torch.empty((3, M, Nn), device=x3.device, dtype=torch.bfloat16) if want_split else None
    if residual is not None and residual.numel() != M * Nn:
        raise IcapError("linear_x3dl: residual has the wrong size")
    span = _Span("linear_x3dl", (M, K, Nn, residual is not None, want_f32, want_split))
    N.check(N.load().icap_linear_x3dl_f32(_ck(x3, "x3", torch.bfloat16), _ck(w3, "w3", torch.bfloat16), _ck(scale, "scale"), _ck(bias, "bias"),
                                          _ck(residual, "residual"), _ck(y, "y"), _ck(y3, "y3", torch.bfloat16), M, K, Nn, act, cfg,
                                          _stream()), "linear_x3dl")
    span.done()
    return y, y3


def linear_x3dl_argmax(x3, w3, bias=None, scale=None, want_logits=False):
    """Vocabulary projection from pre-split operands with fused arg-max partials per 128-column tile
    (icap_linear_x3dl_argmax_f32) -> (part_val [M, P], part_idx [M, P], logits or None)."""
    M, K = x3.shape[1], x3.shape[2]
    Nn = w3.shape[1]
    lib = N.load()
    P = lib.icap_linear_bf16_argmax_parts(Nn)
    pv = torch.empty((M, P), device=x3.device, dtype=torch.float32)
    pi = torch.empty((M, P), device=x3.device, dtype=torch.int32)
    logits = torch.empty((M, Nn), device=x3.device, dtype=torch.float32) if want_logits else None
    span = _Span("linear_x3dl", (M, K, Nn, False, want_logits, False))
    N.check(lib.icap_linear_x3dl_argmax_f32(_ck(x3, "x3", torch.bfloat16), _ck(w3, "w3", torch.bfloat16), _ck(scale, "scale"),
                                            _ck(bias, "bias"), _ck(logits, "logits"), _ck(pv, "pv"), _ck(pi, "pi", torch.int32), M, K, Nn,
                                            _stream()), "linear_x3dl_argmax")
    span.done()
    return pv, pi, logits


def linear_bf16(x, w_bf16, bias=None, scale=None, act=ACT_NONE, residual=None, gate=None, hw=1, out=None, cfg=-1):
    """y[M,N] = act((x[M,K] (* gate[m // hw]) @ w[N,K]^T) * scale + bias (+ residual)) with bf16 products, fp32
    accumulation; w_bf16 from to_bf16(); x fp32 or bf16 (the bf16 result of a fused kernel), everything else fp32."""
    M, K = x.shape
    Nn, K2 = w_bf16.shape[0], w_bf16.numel() // w_bf16.shape[0]
    if K != K2 or w_bf16.dtype != torch.bfloat16:
        raise IcapError(f"linear_bf16: x has K={K}, w must be bf16 [N, {K}]")
    if out is None:
        out = torch.empty((M, Nn), device=x.device, dtype=torch.float32)
    if residual is not None and residual.numel() != M * Nn:
        raise IcapError("linear_bf16: residual has the wrong size")
    if gate is not None and (gate.shape[-1] != K or gate.shape[0] * hw != M):
        raise IcapError("linear_bf16: gate must be [M // hw, K]")
    x16 = x.dtype == torch.bfloat16
    span = _Span("linear_bf16", (M, K, Nn, residual is not None, gate is not None, x16))
    N.check(N.load().icap_linear_bf16_f32(_ck(x, "x", torch.bfloat16 if x16 else torch.float32), int(x16),
                                          _ck(w_bf16, "w", torch.bfloat16), _ck(scale, "scale"), _ck(bias, "bias"),
                                          _ck(residual, "residual"), _ck(gate, "gate"), hw, _ck(out, "out"), M, K, Nn, act, cfg,
                                          _stream()), "linear_bf16")
    span.done()
    return out


def linear_bf16_argmax(x, w_bf16, bias=None, scale=None, want_logits=False):
    """bf16-product vocabulary projection with fused arg-max partials -> (part_val [M,P], part_idx [M,P], logits or None)."""
    M, K = x.shape
    Nn = w_bf16.shape[0]
    lib = N.load()
    P = lib.icap_linear_bf16_argmax_parts(Nn)
    pv = torch.empty((M, P), device=x.device, dtype=torch.float32)
    pi = torch.empty((M, P), device=x.device, dtype=torch.int32)
    logits = torch.empty((M, Nn), device=x.device, dtype=torch.float32) if want_logits else None
    span = _Span("linear_bf16", (M, K, Nn, False, False))
    N.check(lib.icap_linear_bf16_argmax_f32(_ck(x, "x"), _ck(w_bf16, "w", torch.bfloat16), _ck(scale, "scale"), _ck(bias, "bias"),
                                            _ck(logits, "logits"), _ck(pv, "pv"), _ck(pi, "pi", torch.int32), M, K, Nn, _stream()),
            "linear_bf16_argmax")
    span.done()
    return pv, pi, logits


def topk_sample(logits, k=10, seed=0, offset=0, ids_out=None, want_topk=True):
    """softmax -> top-k -> one multinomial draw per row (the reference's generate_caption head).
    logits [rows, V] (rows may be strided) -> (ids [rows] int64, probs [rows, k], idx [rows, k] int32).
    ids_out: optional int64 column view (stride = its stride(0)) that receives the drawn tokens."""
    if not isinstance(logits, torch.Tensor) or not logits.is_cuda or logits.dtype != torch.float32 or logits.stride(1) != 1:
        raise IcapError("topk_sample: logits must be a float32 tensor on the HIP device with contiguous rows")
    rows, V = logits.shape
    dev = logits.device
    ids = ids_out if ids_out is not None else torch.empty(rows, device=dev, dtype=torch.int64)
    if ids.dtype != torch.int64 or ids.dim() != 1 or ids.shape[0] != rows:
        raise IcapError("topk_sample: ids_out must be an int64 vector with one entry per row")
    probs = torch.empty(rows, k, device=dev, dtype=torch.float32) if want_topk else None
    idx = torch.empty(rows, k, device=dev, dtype=torch.int32) if want_topk else None
    N.check(N.load().icap_topk_sample_f32(logits.data_ptr(), logits.stride(0), rows, V, k, int(seed) & (2 ** 64 - 1),
                                          int(offset) & (2 ** 64 - 1), _ck(probs, "probs"), _ck(idx, "idx", torch.int32),
                                          ids.data_ptr(), ids.stride(0), _stream()), "topk_sample")
    return ids, probs, idx


def linear_argmax(x, w, bias=None, scale=None, want_logits=False):
    """Vocabulary projection with fused arg-max partials -> (part_val [M,P], part_idx [M,P], logits or None)."""
    M, K = x.shape
    Nn = w.shape[0]
    lib = N.load()
    P = lib.icap_linear_argmax_parts(Nn)
    pv = torch.empty((M, P), device=x.device, dtype=torch.float32)
    pi = torch.empty((M, P), device=x.device, dtype=torch.int32)
    y = torch.empty((M, Nn), device=x.device, dtype=torch.float32) if want_logits else None
    span = _Span("linear", (M, K, Nn, False))
    N.check(lib.icap_linear_argmax_f32(_ck(x, "x"), _ck(w, "w"), _ck(scale, "scale"), _ck(bias, "bias"), _ck(y, "y"),
                                       _ck(pv, "pv"), _ck(pi, "pi", torch.int32), M, K, Nn, _stream()), "linear_argmax")
    span.done()
    return pv, pi, y


def next_token(pv, pi, ids, step, table=None, add=None, gamma=None, beta=None, eps=1e-5):
    """ids[:, step] = argmax from partials; returns LN(table[id] + add) (or the plain gathered rows, or None)."""
    rows, P = pv.shape
    out = None
    E = 0
    if table is not None:
        E = table.shape[1]
        out = torch.empty((rows, E), device=pv.device, dtype=torch.float32)
    ids_ptr = _ck(ids, "ids", torch.int64) + 8 * step
    N.check(N.load().icap_next_token_f32(_ck(pv, "pv"), _ck(pi, "pi", torch.int32), P, ids_ptr, ids.shape[1], _ck(table, "table"),
                                         _ck(add, "add"), _ck(gamma, "gamma"), _ck(beta, "beta"), eps, _ck(out, "out"), max(E, 1),
                                         rows, max(E, 1), _stream()), "next_token")
    return out


def mha_core(q, k, v, heads):
    """q [Lq,N,E]; k, v [Lk,Nk,E] with Nk in (N, 1)."""
    Lq, Nn, E = q.shape
    Lk, Nk, E2 = k.shape
    if E2 != E or tuple(v.shape) != tuple(k.shape):
        raise IcapError("mha_core: shape mismatch")
    out = torch.empty_like(q)
    N.check(N.load().icap_mha_core_f32(_ck(q, "q"), _ck(k, "k"), _ck(v, "v"), _ck(out, "out"), Lq, Lk, Nn, Nk, E, heads,
                                       _stream()), "mha_core")
    return out


def bahdanau_attention(u, w, a_w, a_b: float, feat, want_alpha=True, alpha_out=None, ctx_out=None):
    """alpha_out [B, B] / ctx_out [B, E]: optional contiguous destinations (a training loop's per-step slices)."""
    B, A = u.shape
    E = feat.shape[-1]
    if tuple(w.shape) != (B, A) or a_w.numel() != A or feat.numel() != B * E:
        raise IcapError("bahdanau_attention: shape mismatch")
    if (alpha_out is not None and tuple(alpha_out.shape) != (B, B)) or (ctx_out is not None and tuple(ctx_out.shape) != (B, E)):
        raise IcapError("bahdanau_attention: alpha_out must be [B, B] and ctx_out [B, E]")
    alpha = alpha_out if alpha_out is not None else (torch.empty((B, B), device=u.device, dtype=torch.float32) if want_alpha else None)
    ctx = ctx_out if ctx_out is not None else torch.empty((B, E), device=u.device, dtype=torch.float32)
    N.check(N.load().icap_bahdanau_attention_f32(_ck(u, "u"), _ck(w, "w"), _ck(a_w, "a_w"), float(a_b), _ck(feat, "feat"),
                                                 _ck(alpha, "alpha"), _ck(ctx, "ctx"), B, A, E, _stream()), "bahdanau_attention")
    return alpha, ctx


def lstm_pointwise(gates, c_in, h_out=None, c_out=None):
    """h_out / c_out [B, H]: optional contiguous destinations (a training loop's per-step slices)."""
    B, H4 = gates.shape
    H = H4 // 4
    if (h_out is not None and tuple(h_out.shape) != (B, H)) or (c_out is not None and tuple(c_out.shape) != (B, H)):
        raise IcapError("lstm_pointwise: h_out and c_out must be [B, H]")
    h = h_out if h_out is not None else torch.empty((B, H), device=gates.device, dtype=torch.float32)
    c = c_out if c_out is not None else torch.empty((B, H), device=gates.device, dtype=torch.float32)
    N.check(N.load().icap_lstm_pointwise_f32(_ck(gates, "gates"), _ck(c_in, "c_in"), _ck(h, "h"), _ck(c, "c"), B, H, _stream()),
            "lstm_pointwise")
    return h, c


def gru_pointwise(gi, gh, h_in):
    B, H3 = gi.shape
    H = H3 // 3
    h = torch.empty((B, H), device=gi.device, dtype=torch.float32)
    N.check(N.load().icap_gru_pointwise_f32(_ck(gi, "gi"), _ck(gh, "gh"), _ck(h_in, "h_in"), _ck(h, "h"), B, H, _stream()),
            "gru_pointwise")
    return h


def gemv(W, x, bias=None, scale=None, act=ACT_NONE, want_partials=False):
    """x [NB<=8, K], W [N, K] -> y [NB, N] (and arg-max partials when asked)."""
    NB, K = x.shape
    Nn = W.shape[0]
    lib = N.load()
    y = torch.empty((NB, Nn), device=x.device, dtype=torch.float32)
    pv = pi = None
    if want_partials:
        P = lib.icap_gemv_blocks(Nn)
        pv = torch.empty((NB, P), device=x.device, dtype=torch.float32)
        pi = torch.empty((NB, P), device=x.device, dtype=torch.int32)
    N.check(lib.icap_gemv_f32(_ck(W, "W"), _ck(scale, "scale"), _ck(bias, "bias"), _ck(x, "x"), K, _ck(y, "y"), Nn, NB, Nn, K,
                              act, _ck(pv, "pv"), _ck(pi, "pi", torch.int32), _stream()), "gemv")
    return (y, pv, pi) if want_partials else y


def lstm_cell(w_ih, w_hh, b_ih, b_hh, x, h, c):
    NB, I = x.shape
    H = h.shape[1]
    ho, co = torch.empty_like(h), torch.empty_like(c)
    N.check(N.load().icap_lstm_cell_f32(_ck(w_ih, "w_ih"), _ck(w_hh, "w_hh"), _ck(b_ih, "b_ih"), _ck(b_hh, "b_hh"),
                                        _ck(x, "x"), I, _ck(h, "h"), _ck(c, "c"), _ck(ho, "ho"), _ck(co, "co"), NB, I, H,
                                        _stream()), "lstm_cell")
    return ho, co


def gru_cell(w_ih, w_hh, b_ih, b_hh, x, h):
    NB, I = x.shape
    H = h.shape[1]
    ho = torch.empty_like(h)
    N.check(N.load().icap_gru_cell_f32(_ck(w_ih, "w_ih"), _ck(w_hh, "w_hh"), _ck(b_ih, "b_ih"), _ck(b_hh, "b_hh"),
                                       _ck(x, "x"), I, _ck(h, "h"), _ck(ho, "ho"), NB, I, H, _stream()), "gru_cell")
    return ho


# ---------------------------------------------------------------- greedy drivers
def _ptr(t):
    return C.c_void_p(_ck(t, "weight"))


# Small-batch decode streams every weight once per step, so its time follows the bytes.  "bf16": the step-per-launch
# drivers read bf16 copies of the weight MATRICES (made once per weight version, kept in the model's weight dict) and
# keep everything else - tables, biases, inputs, state, sums, logits - in fp32.  None: follow the compute dtype ("bf16"
# compute mode streams bf16 weights, the fp32 modes fp32 ones).
DECODE_WEIGHTS = None


def set_decode_weights(name) -> None:
    global DECODE_WEIGHTS
    if name not in (None, "f32", "bf16"):
        raise IcapError(f"set_decode_weights: unknown dtype {name!r}")
    DECODE_WEIGHTS = name


def decode_weights_bf16() -> bool:
    return (DECODE_WEIGHTS or ("bf16" if COMPUTE_DTYPE == "bf16" else "f32")) == "bf16"


def _w16(wd: dict, key: str, idx=None) -> int:
    """Address of the bf16 copy of weight matrix wd[key] (or wd[key][idx]); the copy is made on first use and lives in wd,
    which is rebuilt whenever the weights change."""
    ck = key + "_w16" if idx is None else f"{key}_w16_{idx}"
    t = wd.get(ck)
    if t is None:
        src = wd[key] if idx is None else wd[key][idx]
        if src.shape[-1] % 8:
            raise IcapError(f"bf16 weight streaming: {key} rows of {src.shape[-1]} are not a multiple of 8")
        t = wd[ck] = to_bf16(src.contiguous())
    return t.data_ptr()


# Whole-caption HIP graphs for the lock-step greedy drivers (opt-in).  icap_greedy_* enqueue every launch of a caption
# without a host round trip, so the launches of one call can be recorded once and replayed: a replay costs one graph launch
# on the host, and dependent kernels inside a graph follow each other faster than stream-ordered launches do.  The recorded
# graph reads its inputs from static buffers (a call copies into them) and writes ids / logits / workspace owned by the
# graph's memory pool; results are returned as copies.  Graphs live inside the prepared-weights dict `wd`, which is rebuilt
# whenever the weights change, so a graph never outlives the weight copies it points to.
DECODE_GRAPH = False
_DECODE_GRAPH_MAX = 8        # graphs kept per prepared-weights dict (distinct batch sizes / step counts)


def set_decode_graph(flag: bool) -> None:
    global DECODE_GRAPH
    DECODE_GRAPH = bool(flag)


def _graphed_caption(wd: dict, key, inputs, run):
    """run(inputs) -> tuple of tensors / None, either directly or - with DECODE_GRAPH on - as a replay of the HIP graph
    recorded for (wd, key) on static copies of `inputs`."""
    if not DECODE_GRAPH or torch.cuda.is_current_stream_capturing():
        return run(inputs)
    cache = wd.setdefault("_caption_graphs", {})
    ent = cache.get(key)
    if ent is None:
        static = [t.clone() for t in inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):            # first run outside the recording: lazy weight copies are made here
            run(static)
        torch.cuda.current_stream().wait_stream(side)
        tune = AUTOTUNE
        set_autotune(False)                      # tuning synchronises; a recording must not
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = run(static)
        finally:
            set_autotune(tune)
        if len(cache) >= _DECODE_GRAPH_MAX:
            cache.pop(next(iter(cache)))
        ent = cache[key] = (graph, static, outs)
    graph, static, outs = ent
    for st, t in zip(static, inputs):
        st.copy_(t)
    graph.replay()
    return tuple(o.clone() if o is not None else None for o in outs)


# Latency form of the recurrent greedy drivers (icap_greedy_bah_fused_f32 / icap_greedy_rnn2_fused_f32): the token-dependent
# input side of a cell is a row of a [V, G H] table made once per weight version, the recurrent side is computed a step ahead
# by the vocabulary-projection launch, and the cell update is redone in every workgroup's prologue - one launch per step for
# the Bahdanau LSTMCell decoder, two for the two-layer LSTM / GRU.  Every workgroup repeats the prologue, so the form pays
# only while few captions are in flight: it is taken for at most DECODE_FUSED_MAX_NB captions per call, fp32 weights.
DECODE_FUSED = True
DECODE_FUSED_MAX_NB = 2


def set_decode_fused(flag: bool) -> None:
    global DECODE_FUSED
    DECODE_FUSED = bool(flag)


def _rnn2_fused_tables(wd: dict) -> dict:
    """Prepared operands of icap_greedy_rnn2_fused_f32, kept in wd (rebuilt with the weights)."""
    t = wd.get("_fused")
    if t is None:
        lstm = wd["is_lstm"]
        in_bias0 = wd["b_ih"][0] + wd["b_hh"][0] if lstm else wd["b_ih"][0]
        t = wd["_fused"] = {
            "in_bias0": in_bias0.contiguous(),
            "g_tab0": linear(wd["in_table"], wd["w_ih"][0], bias=in_bias0),
            "ih1_bias": (wd["b_ih"][1] + wd["b_hh"][1]).contiguous() if lstm else wd["b_ih"][1],
            "hh0_bias": None if lstm else wd["b_hh"][0],
            "hh1_bias": None if lstm else wd["b_hh"][1],
        }
    return t


def greedy_rnn2_fused(wd: dict, x0: torch.Tensor, steps: int, want_logits=False):
    """greedy_rnn for a two-layer LSTM / GRU whose layers are all H wide, two launches per step
    (icap_greedy_rnn2_fused_f32).  x0 [NB, H]: feature_fc of the image feature, the first step's input."""
    lib = N.load()
    NB, I = x0.shape
    H = wd["w_hh"][0].shape[1]
    V = wd["fc_w"].shape[0]
    if len(wd["w_ih"]) != 2 or I != H or wd["w_ih"][1].shape[1] != H:
        raise IcapError("greedy_rnn2_fused: needs two layers of one width")
    t = _rnn2_fused_tables(wd)
    lstm = int(wd["is_lstm"])
    dev = x0.device
    ws_floats = lib.icap_greedy_rnn2_fused_ws_floats(NB, lstm, H, V)

    def run(inp):
        g_first = linear(inp[0], wd["w_ih"][0], bias=t["in_bias0"])
        ids = torch.empty((NB, steps), device=dev, dtype=torch.int64)
        logits = torch.empty((steps, NB, V), device=dev, dtype=torch.float32) if want_logits else None
        ws = torch.empty(ws_floats, device=dev, dtype=torch.float32)
        N.check(lib.icap_greedy_rnn2_fused_f32(lstm, _ck(t["g_tab0"], "g_tab0"), _ck(g_first, "g_first"), _ck(wd["w_hh"][0], "w_hh0"),
                                               _ck(t["hh0_bias"], "hh0_bias"), _ck(wd["w_ih"][1], "w_ih1"), _ck(t["ih1_bias"], "ih1_bias"),
                                               _ck(wd["w_hh"][1], "w_hh1"), _ck(t["hh1_bias"], "hh1_bias"), _ck(wd["fc_w"], "fc_w"),
                                               _ck(wd["fc_b"], "fc_b"), _ck(ids, "ids", torch.int64), _ck(logits, "logits"), _ck(ws, "ws"),
                                               NB, H, V, steps, _stream()), "greedy_rnn2_fused")
        return ids, logits
    return _graphed_caption(wd, ("rnn2f", NB, steps, bool(want_logits)), (x0,), run)


def greedy_rnn2_persist_max_nb(is_lstm: bool, H: int, V: int) -> int:
    """Captions per call of the persistent whole-caption two-layer LSTM / GRU decoder for this shape; 0 = not supported."""
    return N.load().icap_greedy_rnn2_persist_max_nb(int(is_lstm), H, V)


def greedy_rnn2_persist(wd: dict, x0: torch.Tensor, steps: int, want_logits=False, check=True):
    """greedy_rnn2_fused as ONE persistent launch with W_ih1, both W_hh and the vocabulary projection resident in LDS
    (icap_greedy_rnn2_persist_f32).  The kernel needs all its workgroups resident at once (a plain launch on an otherwise
    idle device); a launch that could not make progress gives up after a bounded spin and leaves -1 in every id.  check=True
    (the default) reads the ids back - one host synchronisation - and raises IcapError for such a launch."""
    lib = N.load()
    NB, I = x0.shape
    H = wd["w_hh"][0].shape[1]
    V = wd["fc_w"].shape[0]
    if len(wd["w_ih"]) != 2 or I != H or wd["w_ih"][1].shape[1] != H:
        raise IcapError("greedy_rnn2_persist: needs two layers of one width")
    t = _rnn2_fused_tables(wd)
    lstm = int(wd["is_lstm"])
    dev = x0.device
    g_first = linear(x0, wd["w_ih"][0], bias=t["in_bias0"])
    ids = torch.empty((NB, steps), device=dev, dtype=torch.int64)
    logits = torch.empty((steps, NB, V), device=dev, dtype=torch.float32) if want_logits else None
    ws = torch.empty((lib.icap_greedy_rnn2_persist_ws_bytes(NB, lstm, H, V) + 7) // 8, device=dev, dtype=torch.int64)
    N.check(lib.icap_greedy_rnn2_persist_f32(lstm, _ck(t["g_tab0"], "g_tab0"), _ck(g_first, "g_first"), _ck(wd["w_hh"][0], "w_hh0"),
                                             _ck(t["hh0_bias"], "hh0_bias"), _ck(wd["w_ih"][1], "w_ih1"), _ck(t["ih1_bias"], "ih1_bias"),
                                             _ck(wd["w_hh"][1], "w_hh1"), _ck(t["hh1_bias"], "hh1_bias"), _ck(wd["fc_w"], "fc_w"),
                                             _ck(wd["fc_b"], "fc_b"), _ck(ids, "ids", torch.int64), _ck(logits, "logits"), ws.data_ptr(),
                                             NB, H, V, steps, _stream()), "greedy_rnn2_persist")
    if check and bool((ids < 0).any().item()):
        raise IcapError("greedy_rnn2_persist: the persistent launch could not make progress (its workgroups were not all resident); "
                        "use greedy_rnn")
    return ids, logits


def _rnn_fused_ok(wd: dict, NB: int, I: int) -> bool:
    H = wd["w_hh"][0].shape[1]
    return (DECODE_FUSED and NB <= DECODE_FUSED_MAX_NB and len(wd["w_ih"]) == 2 and I == H and H % 4 == 0
            and wd["w_ih"][1].shape[1] == H)


def greedy_rnn(wd: dict, x0: torch.Tensor, steps: int, want_logits=False, ids=None, logits=None, static_ws=False, w16=None):
    """wd: dict with is_lstm, w_ih/w_hh/b_ih/b_hh lists, fc_w, fc_b, in_table.  x0 [NB, I].
    static_ws: the entry point whose workspace lives inside the library (nothing is allocated when `ids` - and `logits`
    if wanted - are passed in); calls that overlap on one GPU share that workspace and must be serialised.
    w16: stream bf16 copies of the weight matrices (icap_greedy_rnn_w16_f32); None follows decode_weights_bf16().
    One or two captions of a two-layer decoder with fp32 weights take the two-launches-per-step form (greedy_rnn2_fused)
    unless the caller passes buffers or set_decode_fused(False)."""
    w16 = decode_weights_bf16() if w16 is None else w16
    if w16 and static_ws:
        raise IcapError("greedy_rnn: the built-in-workspace entry point streams fp32 weights only")
    lib = N.load()
    NB, I = x0.shape
    if not w16 and not static_ws and ids is None and logits is None and _rnn_fused_ok(wd, NB, I):
        return greedy_rnn2_fused(wd, x0, steps, want_logits)
    L = len(wd["w_ih"])
    H = wd["w_hh"][0].shape[1]
    V = wd["fc_w"].shape[0]
    w = N.RnnWeights()
    w.layers, w.H, w.I, w.V, w.is_lstm = L, H, I, V, int(wd["is_lstm"])
    for l in range(L):
        w.w_ih[l] = _w16(wd, "w_ih", l) if w16 else _ck(wd["w_ih"][l], "w_ih")
        w.w_hh[l] = _w16(wd, "w_hh", l) if w16 else _ck(wd["w_hh"][l], "w_hh")
        w.b_ih[l] = _ck(wd["b_ih"][l], "b_ih"); w.b_hh[l] = _ck(wd["b_hh"][l], "b_hh")
    w.fc_w = _w16(wd, "fc_w") if w16 else _ck(wd["fc_w"], "fc_w")
    w.fc_b, w.in_table = _ck(wd["fc_b"], "fc_b"), _ck(wd["in_table"], "in_table")
    dev = x0.device
    user_buffers = ids is not None or logits is not None
    if ids is None:
        ids = torch.empty((NB, steps), device=dev, dtype=torch.int64)
    elif tuple(ids.shape) != (NB, steps):
        raise IcapError("greedy_rnn: ids must be [NB, steps]")
    if want_logits and logits is None:
        logits = torch.empty((steps, NB, V), device=dev, dtype=torch.float32)
    elif logits is not None and tuple(logits.shape) != (steps, NB, V):
        raise IcapError("greedy_rnn: logits must be [steps, NB, V]")
    if static_ws:
        N.check(lib.icap_greedy_rnn_static_f32(C.byref(w), _ck(x0, "x0"), _ck(ids, "ids", torch.int64), _ck(logits, "logits"),
                                               NB, steps, _stream()), "greedy_rnn_static")
        return ids, logits
    fn = lib.icap_greedy_rnn_w16_f32 if w16 else lib.icap_greedy_rnn_f32
    ws_floats = lib.icap_greedy_rnn_ws_floats(NB, L, H, I, V)
    if user_buffers or not DECODE_GRAPH:
        ws = torch.empty(ws_floats, device=dev, dtype=torch.float32)
        N.check(fn(C.byref(w), _ck(x0, "x0"), _ck(ids, "ids", torch.int64), _ck(logits, "logits"), _ck(ws, "ws"), NB, steps, _stream()),
                "greedy_rnn")
        return ids, logits

    def run(inp):        # buffers allocated here: inside the recording they belong to the graph's pool
        i = torch.empty((NB, steps), device=dev, dtype=torch.int64)
        lg = torch.empty((steps, NB, V), device=dev, dtype=torch.float32) if want_logits else None
        ws = torch.empty(ws_floats, device=dev, dtype=torch.float32)
        N.check(fn(C.byref(w), _ck(inp[0], "x0"), _ck(i, "ids", torch.int64), _ck(lg, "logits"), _ck(ws, "ws"), NB, steps, _stream()),
                "greedy_rnn")
        return i, lg
    return _graphed_caption(wd, ("rnn", NB, I, steps, bool(want_logits), bool(w16)), (x0,), run)


def greedy_tfm(wd: dict, ca: torch.Tensor, start_ids: torch.Tensor, steps: int, want_logits=False, w16=None):
    """w16: stream bf16 copies of fc1 / fc2 / output_fc (icap_greedy_tfm_w16_f32); None follows decode_weights_bf16()."""
    lib = N.load()
    w16 = decode_weights_bf16() if w16 is None else w16
    NB, E = ca.shape
    FF = wd["fc1_w"].shape[0]
    V = wd["out_w"].shape[0]
    w = N.TfmWeights()
    w.E, w.FF, w.V = E, FF, V
    for k in ("emb", "ln2_g", "ln2_b", "ln3_g", "ln3_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "out_w", "out_b"):
        setattr(w, k, _w16(wd, k) if w16 and k in ("fc1_w", "fc2_w", "out_w") else _ck(wd[k], k))
    dev = ca.device
    fn = lib.icap_greedy_tfm_w16_f32 if w16 else lib.icap_greedy_tfm_f32
    ws_floats = lib.icap_greedy_tfm_ws_floats(NB, E, FF, V)

    def run(inp):
        ids = torch.empty((NB, steps), device=dev, dtype=torch.int64)
        logits = torch.empty((steps, NB, V), device=dev, dtype=torch.float32) if want_logits else None
        ws = torch.empty(ws_floats, device=dev, dtype=torch.float32)
        N.check(fn(C.byref(w), _ck(inp[0], "ca"), _ck(inp[1], "start_ids", torch.int64), _ck(ids, "ids", torch.int64),
                   _ck(logits, "logits"), _ck(ws, "ws"), NB, steps, _stream()), "greedy_tfm")
        return ids, logits
    return _graphed_caption(wd, ("tfm", NB, steps, bool(want_logits), bool(w16)), (ca, start_ids), run)


def table_walk(nxt, start, steps, logits_tab=None, V=None):
    """nxt int64 [V, B] next-token table, start int64 [B] -> ids [B, steps] (and logits [steps, B, V] gathered from
    logits_tab [V*B, ld] when given)."""
    Vt, B = nxt.shape
    V = Vt if V is None else V
    ids = torch.empty((B, steps), device=nxt.device, dtype=torch.int64)
    logits = torch.empty((steps, B, V), device=nxt.device, dtype=torch.float32) if logits_tab is not None else None
    N.check(N.load().icap_table_walk(_ck(nxt, "next", torch.int64), _ck(start, "start", torch.int64), _ck(ids, "ids", torch.int64),
                                     _ck(logits_tab, "logits_tab"), logits_tab.shape[1] if logits_tab is not None else 0,
                                     _ck(logits, "logits"), V, B, steps, _stream()), "table_walk")
    return ids, logits


def greedy_tfm_persist_max_nb(E: int, FF: int, V: int) -> int:
    """Captions per call of the persistent whole-caption decoder for this shape; 0 = not supported."""
    return N.load().icap_greedy_tfm_persist_max_nb(E, FF, V)


def greedy_tfm_persist(wd: dict, ca: torch.Tensor, start_ids: torch.Tensor, steps: int, want_logits=False, check=True):
    """greedy_tfm as one persistent launch with LDS-resident weights (icap_greedy_tfm_persist_f32).  The kernel needs all
    its workgroups resident at once (a plain launch: a kernel running beside it on another stream, or a CU-masked device,
    breaks that); a launch that could not make progress gives up after a bounded spin and leaves -1 in every id.  With
    check=True (the default) the ids are read back - one host synchronisation - and such a launch raises IcapError
    instead of returning -1 as if it were a caption; check=False leaves the test to the caller."""
    lib = N.load()
    NB, E = ca.shape
    FF = wd["fc1_w"].shape[0]
    V = wd["out_w"].shape[0]
    w = N.TfmWeights()
    w.E, w.FF, w.V = E, FF, V
    for k in ("emb", "ln2_g", "ln2_b", "ln3_g", "ln3_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "out_w", "out_b"):
        setattr(w, k, _ck(wd[k], k))
    dev = ca.device
    ids = torch.empty((NB, steps), device=dev, dtype=torch.int64)
    logits = torch.empty((steps, NB, V), device=dev, dtype=torch.float32) if want_logits else None
    ws = torch.empty((lib.icap_greedy_tfm_persist_ws_bytes(NB, E, FF, V) + 7) // 8, device=dev, dtype=torch.int64)
    N.check(lib.icap_greedy_tfm_persist_f32(C.byref(w), _ck(ca, "ca"), _ck(start_ids, "start_ids", torch.int64),
                                            _ck(ids, "ids", torch.int64), _ck(logits, "logits"), ws.data_ptr(), NB, steps, _stream()),
            "greedy_tfm_persist")
    if check and bool((ids < 0).any().item()):
        raise IcapError("greedy_tfm_persist: the persistent launch could not make progress (its workgroups were not all resident); "
                        "use greedy_tfm")
    return ids, logits


def greedy_bah_fused(wd: dict, context: torch.Tensor, start_ids: torch.Tensor, steps: int, want_logits=False):
    """greedy_bah with one launch per step (icap_greedy_bah_fused_f32): the embedding half of W_ih becomes the table
    g_tab[v] = W_ih[:, :E] . emb[v] (kept in wd), the context half g_ctx = W_ih[:, E:] . context + b_ih + b_hh is made per
    call, W_hh . h rides on the vocabulary-projection launch."""
    lib = N.load()
    NB, ENC = context.shape
    V, E = wd["emb"].shape
    H = wd["w_hh"].shape[1]
    if "g_tab" not in wd:
        wd["g_tab"] = linear(wd["emb"], wd["w_ih_e"])
        wd["b_gates"] = (wd["b_ih"] + wd["b_hh"]).contiguous()
    dev = context.device
    ws_floats = lib.icap_greedy_bah_fused_ws_floats(NB, H, V)

    def run(inp):
        g_ctx = linear(inp[0], wd["w_ih_c"], bias=wd["b_gates"])
        ids = torch.empty((NB, steps), device=dev, dtype=torch.int64)
        logits = torch.empty((steps, NB, V), device=dev, dtype=torch.float32) if want_logits else None
        ws = torch.empty(ws_floats, device=dev, dtype=torch.float32)
        N.check(lib.icap_greedy_bah_fused_f32(_ck(wd["g_tab"], "g_tab"), _ck(g_ctx, "g_ctx"), _ck(wd["w_hh"], "w_hh"), _ck(wd["fc_w"], "fc_w"),
                                              _ck(wd["fc_scale"], "fc_scale"), _ck(wd["fc_bias"], "fc_bias"),
                                              _ck(inp[1], "start_ids", torch.int64), _ck(ids, "ids", torch.int64), _ck(logits, "logits"),
                                              _ck(ws, "ws"), NB, H, V, steps, _stream()), "greedy_bah_fused")
        return ids, logits
    return _graphed_caption(wd, ("bahf", NB, ENC, steps, bool(want_logits)), (context, start_ids), run)


def greedy_bah_persist_max_nb(H: int, V: int) -> int:
    """Captions per call of the persistent whole-caption LSTMCell decoder for this shape; 0 = not supported."""
    return N.load().icap_greedy_bah_persist_max_nb(H, V)


def greedy_bah_persist(wd: dict, context: torch.Tensor, start_ids: torch.Tensor, steps: int, want_logits=False, check=True):
    """greedy_bah_fused as ONE persistent launch with the vocabulary projection and W_hh resident in LDS
    (icap_greedy_bah_persist_f32).  The kernel needs all its workgroups resident at once (a plain launch on an otherwise idle
    device); a launch that could not make progress gives up after a bounded spin and leaves -1 in every id.  check=True (the
    default) reads the ids back - one host synchronisation - and raises IcapError for such a launch."""
    lib = N.load()
    NB, ENC = context.shape
    V, E = wd["emb"].shape
    H = wd["w_hh"].shape[1]
    if "g_tab" not in wd:
        wd["g_tab"] = linear(wd["emb"], wd["w_ih_e"])
        wd["b_gates"] = (wd["b_ih"] + wd["b_hh"]).contiguous()
    dev = context.device
    g_ctx = linear(context, wd["w_ih_c"], bias=wd["b_gates"])
    ids = torch.empty((NB, steps), device=dev, dtype=torch.int64)
    logits = torch.empty((steps, NB, V), device=dev, dtype=torch.float32) if want_logits else None
    ws = torch.empty((lib.icap_greedy_bah_persist_ws_bytes(NB, H, V) + 7) // 8, device=dev, dtype=torch.int64)
    N.check(lib.icap_greedy_bah_persist_f32(_ck(wd["g_tab"], "g_tab"), _ck(g_ctx, "g_ctx"), _ck(wd["w_hh"], "w_hh"), _ck(wd["fc_w"], "fc_w"),
                                            _ck(wd["fc_scale"], "fc_scale"), _ck(wd["fc_bias"], "fc_bias"),
                                            _ck(start_ids, "start_ids", torch.int64), _ck(ids, "ids", torch.int64), _ck(logits, "logits"),
                                            ws.data_ptr(), NB, H, V, steps, _stream()), "greedy_bah_persist")
    if check and bool((ids < 0).any().item()):
        raise IcapError("greedy_bah_persist: the persistent launch could not make progress (its workgroups were not all resident); "
                        "use greedy_bah")
    return ids, logits


def greedy_bah(wd: dict, context: torch.Tensor, start_ids: torch.Tensor, steps: int, want_logits=False, w16=None):
    """w16: stream bf16 copies of the LSTMCell matrices and fc (icap_greedy_bah_w16_f32); None follows decode_weights_bf16().
    One or two captions with fp32 weights take the one-launch-per-step form (greedy_bah_fused) unless set_decode_fused(False)."""
    lib = N.load()
    w16 = decode_weights_bf16() if w16 is None else w16
    NB, ENC = context.shape
    V, E = wd["emb"].shape
    H = wd["w_hh"].shape[1]
    if not w16 and DECODE_FUSED and NB <= DECODE_FUSED_MAX_NB and H % 4 == 0:
        return greedy_bah_fused(wd, context, start_ids, steps, want_logits)
    w = N.BahWeights()
    w.E, w.H, w.V = E, H, V
    for k in ("emb", "w_ih", "w_hh", "b_ih", "b_hh", "fc_w", "fc_scale", "fc_bias"):
        setattr(w, k, _w16(wd, k) if w16 and k in ("w_ih", "w_hh", "fc_w") else _ck(wd[k], k))
    dev = context.device
    fn = lib.icap_greedy_bah_w16_f32 if w16 else lib.icap_greedy_bah_f32
    ws_floats = lib.icap_greedy_bah_ws_floats(NB, E, ENC, H, V)

    def run(inp):
        ids = torch.empty((NB, steps), device=dev, dtype=torch.int64)
        logits = torch.empty((steps, NB, V), device=dev, dtype=torch.float32) if want_logits else None
        ws = torch.empty(ws_floats, device=dev, dtype=torch.float32)
        N.check(fn(C.byref(w), _ck(inp[0], "context"), ENC, _ck(inp[1], "start_ids", torch.int64), _ck(ids, "ids", torch.int64),
                   _ck(logits, "logits"), _ck(ws, "ws"), NB, steps, _stream()), "greedy_bah")
        return ids, logits
    return _graphed_caption(wd, ("bah", NB, ENC, steps, bool(want_logits), bool(w16)), (context, start_ids), run)
