"""Deterministic image preprocessing.

The reference evaluates with ``data.image_transforms`` (data.py:166-173), which applies a random flip,
rotation and contrast jitter even at test time (TransformerTest.py:348), so its inputs are not
reproducible.  This is the deterministic tail of that pipeline: RGB -> 224x224 bilinear resize ->
[0,1] -> ImageNet mean/std, NCHW fp32.

``augment`` / ``draw_augment_params`` are the random head of the pipeline (flip, rotation, contrast jitter) on the GPU
for explicit per-image parameters; ``data.image_transforms(..., augment=True)`` draws them the way torchvision would.

Two routes to the same tensor:

* ``load_image_u8`` + ``to_model_input``: Pillow resizes on the host (what torchvision's ``Resize`` does with a PIL
  image), the normalisation is torch arithmetic.  Used for fixtures.
* ``resize_normalize``: the decoded uint8 pixels go to the GPU as they are and ``icap_resize_normalize_u8_f32``
  produces the normalised tensor, bit-identical to the first route (Pillow's fixed-point triangle filter is
  reproduced tap for tap).  The host only decodes the JPEG.
"""
from __future__ import annotations

import functools
import math

import numpy as np
import torch

from . import _native as N
from .synthetic import normalize_u8

IMAGENET_MEAN = (0.485, 0.456, 0.406)     # data.py:172
IMAGENET_STD = (0.229, 0.224, 0.225)
_PRECISION_BITS = 32 - 8 - 2              # Pillow's 8-bit resampler keeps taps in 22-bit fixed point


def load_image_u8(path: str, size: int = 224) -> np.ndarray:
    from PIL import Image
    img = Image.open(path).convert("RGB").resize((size, size), Image.BILINEAR)
    return np.asarray(img, dtype=np.uint8)


def decode_image_u8(path: str) -> np.ndarray:
    """JPEG/PNG file -> uint8 [H, W, 3] at its own size (data.py:56: ``Image.open(path).convert("RGB")``)."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)


def to_model_input(u8_nhwc) -> torch.Tensor:
    """uint8 [N, H, W, 3] (numpy or torch) -> normalised fp32 [N, 3, H, W]."""
    t = torch.as_tensor(np.asarray(u8_nhwc))
    if t.dim() == 3:
        t = t.unsqueeze(0)
    return normalize_u8(t)


@functools.lru_cache(maxsize=256)
def bilinear_taps(in_size: int, out_size: int):
    """Filter taps of Pillow's BILINEAR resampler for one axis: (taps int32 [out, ksize], bounds int32 [out, 2]) with
    bounds[i] = (first input index, tap count).  Follows the order of operations of Pillow's coefficient set-up (double
    precision: window centre (i + 0.5) * scale, support max(scale, 1), triangle weights summed left to right and
    normalised by their sum, then rounded to 22-bit fixed point), because the last bit of a tap can change a pixel."""
    if in_size <= 0 or out_size <= 0:
        raise N.IcapError("bilinear_taps: sizes must be positive")
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / fscale
    taps = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    one = float(1 << _PRECISION_BITS)
    for i in range(out_size):
        center = (i + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        n = hi - lo
        w = []
        total = 0.0
        for x in range(n):
            a = abs((x + lo - center + 0.5) * inv)
            v = 1.0 - a if a < 1.0 else 0.0
            w.append(v)
            total += v
        for x in range(n):
            v = w[x] / total if total != 0.0 else w[x]
            taps[i, x] = int(0.5 + v * one)
        bounds[i] = (lo, n)
    assert int(bounds[:, 0].min()) >= 0 and int((bounds[:, 0] + bounds[:, 1]).max()) <= in_size and int(bounds[:, 1].max()) <= ksize
    return taps, bounds


_DEVICE_TABLES: dict = {}


def _device_tables(in_size: int, out_size: int, dev):
    key = (in_size, out_size, str(dev))
    t = _DEVICE_TABLES.get(key)
    if t is None:
        taps, bounds = bilinear_taps(in_size, out_size)
        t = _DEVICE_TABLES[key] = (torch.from_numpy(taps).to(dev), torch.from_numpy(bounds).to(dev), taps.shape[1])
    return t


def _mean_std(mean, std, dev):
    key = ("mean_std", mean, std, str(dev))
    t = _DEVICE_TABLES.get(key)
    if t is None:
        t = _DEVICE_TABLES[key] = torch.tensor(list(mean) + list(std), dtype=torch.float32).to(dev)
    return t


def resize_normalize(u8_nhwc, size: int = 224, layout: str = "nchw", mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """uint8 [B, H, W, 3] (or [H, W, 3]) decoded RGB pixels, all of one size, on the HIP device (a numpy array / CPU
    tensor is copied there) -> fp32 [B, 3, size, size] (``layout="nchw"``, the reference's tensor) or [B, size, size, 4]
    with a zero fourth channel (``"nhwc4"``): Resize((size, size)) + ToTensor + Normalize of data.py:170-172."""
    from . import ops
    t = torch.as_tensor(np.asarray(u8_nhwc)) if not isinstance(u8_nhwc, torch.Tensor) else u8_nhwc
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3:
        raise N.IcapError("resize_normalize: expected uint8 [B, H, W, 3]")
    if layout not in ("nchw", "nhwc4"):
        raise N.IcapError("resize_normalize: layout must be 'nchw' or 'nhwc4'")
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise N.IcapError("resize_normalize: no HIP device; the MI355X path has no CPU fallback")
        t = t.cuda()
    t = t.contiguous()
    B, H, W, _ = t.shape
    dev = t.device
    kx, bx, ksx = _device_tables(W, size, dev)
    ky, by, ksy = _device_tables(H, size, dev)
    ms = _mean_std(tuple(mean), tuple(std), dev)
    lib = N.load()
    tmp = torch.empty(lib.icap_resize_normalize_ws_bytes(B, H, size), device=dev, dtype=torch.uint8)
    out = torch.empty((B, 3, size, size) if layout == "nchw" else (B, size, size, 4), device=dev, dtype=torch.float32)
    span = ops._Span("resize_normalize", (B, H, W, size, layout))
    N.check(lib.icap_resize_normalize_u8_f32(t.data_ptr(), B, H, W, kx.data_ptr(), bx.data_ptr(), ksx, ky.data_ptr(), by.data_ptr(),
                                             ksy, size, size, ms.data_ptr(), tmp.data_ptr(), out.data_ptr(),
                                             0 if layout == "nchw" else 1, ops._stream()), "resize_normalize")
    span.done()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The random head of the reference's transform (data.py:167-169, datachar.py:125-127): RandomHorizontalFlip(0.5),
# RandomRotation(20), ColorJitter(contrast=0.3) on the decoded pixels, before the resize.


def rotation_coeffs(w: int, h: int, angle: float):
    """The 16.16 fixed-point coefficients (a0 .. a5) of Pillow's NEAREST affine pass for ``Image.rotate(angle)`` on a
    w x h image (what torchvision's RandomRotation calls): source x = (a2 + x*a0 + y*a1) >> 16, source y =
    (a5 + x*a3 + y*a4) >> 16.  Pillow's order of operations is kept (angle modulo 360, cos / sin rounded to 15 decimals,
    centre (w/2, h/2), half-pixel offset folded into the constants, floor(v * 65536 + 0.5)); its exact shortcuts (0 and
    180 degrees, 90 / 270 for square images) are expressed as exact integer maps."""
    one, half = 65536, 32768
    angle = angle % 360.0
    if angle == 0:
        return (one, 0, half, 0, one, half)
    if angle == 180:
        return (-one, 0, w * one - half, 0, -one, h * one - half)
    if angle == 90 and w == h:                     # counter-clockwise: out[y][x] = in[x][w - 1 - y]
        return (0, -one, w * one - half, one, 0, half)
    if angle == 270 and w == h:
        return (0, one, half, -one, 0, h * one - half)
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2] + cx
    m[5] = m[3] * -cx + m[4] * -cy + m[5] + cy

    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))
    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def draw_augment_params(n: int, p_flip: float = 0.5, degrees: float = 20.0, contrast: float = 0.3):
    """(flips, angles, factors) for n images, drawn from torch's global CPU generator in the order torchvision's transforms
    draw them for one image after the other: ``torch.rand(1) < p``; ``torch.empty(1).uniform_(-degrees, degrees)``;
    ColorJitter's ``torch.randperm(4)`` (the order of its four adjustments - only contrast is active, the draw is still
    made) then ``torch.empty(1).uniform_(1 - contrast, 1 + contrast)``.  torchvision is absent here, so that this
    reproduces its stream under the same ``torch.manual_seed`` is unverified (parity unpinned); the arithmetic applied for
    given parameters is pinned against Pillow."""
    flips, angles, factors = [], [], []
    for _ in range(n):
        flips.append(bool(torch.rand(1) < p_flip))
        angles.append(float(torch.empty(1).uniform_(-degrees, degrees).item()))
        torch.randperm(4)
        factors.append(float(torch.empty(1).uniform_(max(0.0, 1 - contrast), 1 + contrast)))
    return flips, angles, factors


def augment(u8_nhwc, flips, angles, factors) -> torch.Tensor:
    """uint8 [B, H, W, 3] decoded pixels (numpy / CPU tensor is copied to the device) -> uint8 [B, H, W, 3] on the device:
    image i mirrored if flips[i], rotated by angles[i] degrees counter-clockwise (NEAREST, zero fill, same size), contrast
    scaled by factors[i] about its mean luma - ``icap_augment_u8``, bit-identical to Pillow with the same parameters."""
    from . import ops
    t = torch.as_tensor(np.asarray(u8_nhwc)) if not isinstance(u8_nhwc, torch.Tensor) else u8_nhwc
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3:
        raise N.IcapError("augment: expected uint8 [B, H, W, 3]")
    B, H, W, _ = t.shape
    if not (len(flips) == len(angles) == len(factors) == B):
        raise N.IcapError("augment: one flip / angle / contrast factor per image required")
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise N.IcapError("augment: no HIP device; the MI355X path has no CPU fallback")
        t = t.cuda()
    t = t.contiguous()
    prm = np.zeros((B, 8), np.int32)
    for i in range(B):
        prm[i, 0] = 1 if flips[i] else 0
        prm[i, 1:7] = rotation_coeffs(W, H, float(angles[i]))
    dev = t.device
    prm_d = torch.from_numpy(prm).to(dev)
    fac_d = torch.tensor([float(f) for f in factors], dtype=torch.float32).to(dev)
    lib = N.load()
    ws = torch.empty(max(1, lib.icap_augment_ws_bytes(B) // 8), device=dev, dtype=torch.int64)
    out = torch.empty_like(t)
    span = ops._Span("augment", (B, H, W))
    N.check(lib.icap_augment_u8(t.data_ptr(), B, H, W, prm_d.data_ptr(), fac_d.data_ptr(), ws.data_ptr(), out.data_ptr(),
                                ops._stream()), "augment")
    span.done()
    return out
